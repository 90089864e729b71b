"""Exercises the RCCL (torch.distributed backend "nccl") reduce hook on ONE GPU: a 1-rank process group, a context
told it is rank 0 of 2 so that every statistic goes through the hook.  Numbers are those of half a job; the point
is that the all-reduce on the library's own device buffer works with RCCL."""
import os, sys
sys.path.insert(0, '.')
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import numpy as np, torch, torch.distributed as tdist
from glmmrmcml_amd import api, dist as gdist, synth
torch.cuda.set_device(0)
tdist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
calls = []
inner = gdist.make_reduce_hook()
def hook(user, ptr, n):
    calls.append(n)
    return inner(user, ptr, n)
d = synth.geospatial(600, seed=3)
ctx = api.Context(d["cov"], d["data"], d["eff_range"], d["Z"], d["X"], d["y"], d["family"], d["link"], device=0,
                  rank=0, world=2, reduce=hook)
r = ctx.mcml_full(d["start"], mcnr=True, m=64, maxiter=2, warmup=20, tol=0.0, verbose=False, lambda_=2.0, maxsteps=5,
                  seed=11, chains=64, maxfun=15)
print("beta", r["beta"], "theta", r["theta"], "iters", r["iters"])
print("reduce hook calls:", len(calls), "payload sizes:", sorted(set(calls)))
ctx.close()
# the native path of bench.py on the same 1-rank group: the id travels through torch.distributed, the library makes its
# own communicator, the all-reduce is self-tested on known values (dist.init_native_rccl), then a fit runs through it
ctx = api.Context(d["cov"], d["data"], d["eff_range"], d["Z"], d["X"], d["y"], d["family"], d["link"], device=0)
gdist.init_native_rccl(ctx, 0, 1)
r2 = ctx.mcml_full(d["start"], mcnr=True, m=64, maxiter=2, warmup=20, tol=0.0, verbose=False, lambda_=2.0, maxsteps=5,
                   seed=11, chains=64, maxfun=15)
print("native:", ctx.comm_stats(), "beta", r2["beta"], "theta", r2["theta"])
ctx.close()
tdist.destroy_process_group()
print("nccl rehearsal ok")
