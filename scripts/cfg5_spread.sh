#!/bin/bash
# config 5's process-to-process spread: the sampler under rocprofv3 --kernel-trace in five fresh processes; per-kernel
# average durations side by side, to tell "every kernel scales" (clock / power) from "some kernels" (placement of their arrays)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
for i in 1 2 3 4 5; do
  rocprofv3 --kernel-trace -d /tmp/c5_$i -o k -- python3 scripts/time_cfg.py cfg5 1024 > /tmp/c5_$i.log 2>&1 || exit 1
  python3 scripts/rocpd_stats.py /tmp/c5_$i/k_results.db > /tmp/c5_$i.csv
  grep "hmc warm=50" /tmp/c5_$i.log
done
python3 - <<'P'
import csv
rows = {}
for i in range(1, 6):
    for r in csv.DictReader(open("/tmp/c5_%d.csv" % i)):
        rows.setdefault(r["Name"][:60], {})[i] = (float(r["AverageNs"]) / 1e3, int(r["Calls"]))
print("%-62s %s" % ("kernel (avg us per process)", " ".join("%8d" % i for i in range(1, 6))))
for k, v in sorted(rows.items(), key=lambda kv: -sum(a * c for a, c in kv[1].values()))[:12]:
    print("%-62s %s" % (k, " ".join("%8.1f" % v[i][0] if i in v else "       -" for i in range(1, 6))))
P
