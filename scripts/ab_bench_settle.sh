#!/bin/bash
# the full bench process four times as it is and four times with 0.5 s of idle time between the untimed iteration and the timed
# repetitions of configs 2 / 4 / 5: does config 4's first-repetition stall wait for the process or for the clock?
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for i in 1 2 3 4; do
  for s in "" 0.5; do
    if [ -z "$s" ]; then python3 bench.py --steps 1 --warmup 1 > gpurun_out/settle_tmp.json 2>/dev/null || exit 1
    else GLMMR_BENCH_SETTLE=$s python3 bench.py --steps 1 --warmup 1 > gpurun_out/settle_tmp.json 2>/dev/null || exit 1; fi
    python3 - "$s" <<'P'
import json, sys
d = json.loads(open("gpurun_out/settle_tmp.json").read().strip().splitlines()[-1])
print("settle=%-4s" % (sys.argv[1] or "0"), {k: [round(x, 1) for x in o["ms_per_iter_reps"]] for k, o in d["other_configs"].items()}, flush=True)
P
  done
done
