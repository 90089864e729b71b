"""the bench's other_configs sequence for configs 2 and 4 only, with the per-phase wall clock: to catch a slow first repetition of
config 4 and see which phase it is in"""
import sys, json
sys.path.insert(0, ".")
import torch
import bench
from glmmrmcml_amd import api, synth
stream = torch.cuda.current_stream().cuda_stream
out = bench.other_configs(api, synth, stream)
for k in ("cfg2", "cfg4", "cfg5"):
    o = out[k]
    print(k, [round(x, 1) for x in o["ms_per_iter_reps"]], json.dumps(o["phases_ms_per_iter_reps"]))
