#!/bin/bash
# HBM traffic of a theta-step round of 8 (scripts/time_mvn_batch.py 5000 1024 8): separate FETCH_SIZE / WRITE_SIZE passes,
# total bytes per kernel family over the whole run and per round (the script runs 12 single evaluations + 9 rounds)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_mvnb; mkdir -p $O
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/f -o f -- python3 scripts/time_mvn_batch.py 5000 1024 8 > $O/f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/w -o w -- python3 scripts/time_mvn_batch.py 5000 1024 8 > $O/w.log 2>&1
python3 - <<'PY'
import csv, glob, collections
def load(pat, name):
    tot = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(pat):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != name: continue
            k = row["Kernel_Name"]; gy = int(row.get("Grid_Size_Y", row.get("Grid_Size", 1)) or 1)
            fam = "dgemm_dl" if "dgemm_dl" in k else "leaf" if "potrf_leaf" in k else "build" if "build_dense" in k else "other"
            # batched launches have grid y (or x for the leaf) = 8
            wg = int(row.get("Workgroup_Size", 0) or 0)
            tot[fam][0] += float(row["Counter_Value"]); tot[fam][1] += 1
    return tot
F = load("gpurun_out/pmc_mvnb/f/**/*counter_collection.csv", "FETCH_SIZE") or load("gpurun_out/pmc_mvnb/f/*counter_collection.csv", "FETCH_SIZE")
W = load("gpurun_out/pmc_mvnb/w/**/*counter_collection.csv", "WRITE_SIZE") or load("gpurun_out/pmc_mvnb/w/*counter_collection.csv", "WRITE_SIZE")
print("family, launches, FETCH_SIZE sum (KB units as reported), WRITE_SIZE sum")
for fam in sorted(set(F) | set(W)):
    print(fam, F[fam][1], "%.4g" % F[fam][0], "%.4g" % W[fam][0])
PY
tail -2 $O/f.log
find $O -name "*.csv" -size +1M -delete
