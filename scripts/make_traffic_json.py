"""HBM bytes per launch of the HMC GEMMs from the two rocprofv3 PMC passes of bench.py (FETCH_SIZE, WRITE_SIZE),
in the form bench.py quotes (profiles/r02_hbm_traffic.json).
usage: python scripts/make_traffic_json.py <fetch_dir> <write_dir> <n> <chains> <dense_z 0|1> <build-id> > out.json
gfx950 corrections (MI355X_MICROARCH.md, HBM): FETCH_SIZE (KB) counts wide coalesced reads at half their bytes -> x2;
WRITE_SIZE (KB) is exact."""
import csv, glob, json, sys
from collections import defaultdict


def collect(d, counter):
    acc = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != counter:
                continue
            a = acc[row["Kernel_Name"]]
            a[0] += float(row["Counter_Value"]); a[1] += 1
    return acc


fd, wd, n, chains, dz, build = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
fe, wr = collect(fd, "FETCH_SIZE"), collect(wd, "WRITE_SIZE")
detail = {}
for k in sorted(set(fe) | set(wr)):
    if not any(p in k for p in ("dgemm_band_kernel", "k_band_reduce", "dgemm_dlds", "k_cm_forward", "k_cm_backward", "k_cm_Lrow", "k_cm_Lcol",
                                  "k_cm_logprob", "k_cm_propose", "k_cm_commit")):
        continue
    f, nf = fe.get(k, [0.0, 0]); w, nw = wr.get(k, [0.0, 0])
    detail[k[:120]] = {"launches": nf, "fetch_bytes_per_launch_x2": 2 * 1024 * f / max(nf, 1),
                       "write_bytes_per_launch": 1024 * w / max(nw, 1)}


def avg(pred):
    v = [d["fetch_bytes_per_launch_x2"] + d["write_bytes_per_launch"] for k, d in detail.items() if pred(k)]
    return sum(v) / len(v) if v else None


out = {"n": n, "chains": chains, "dense_z": bool(dz), "build": build,
       "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `python3 bench.py`; FETCH_SIZE x2 "
                 "(gfx950 counts wide coalesced reads at half their bytes), WRITE_SIZE as is; averaged over the forward "
                 "and backward kernels",
       "hbm_bytes_per_launch_banded": avg(lambda k: "dgemm_band_kernel" in k),
       "hbm_bytes_per_launch_dense": avg(lambda k: "dgemm_dlds" in k),
       "hbm_bytes_per_launch_sparse": avg(lambda k: "k_cm_forward" in k or "k_cm_backward" in k),
       "kernels": detail}
print(json.dumps(out, indent=1))
