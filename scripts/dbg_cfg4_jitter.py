import sys, time
sys.path.insert(0, ".")
import numpy as np
import torch
from glmmrmcml_amd import api, synth
mode = sys.argv[1] if len(sys.argv) > 1 else "with2"
stream = torch.cuda.current_stream().cuda_stream
if mode == "with2":
    d2 = synth.geospatial(2000, seed=1)
    with api.Context(d2["cov"], d2["data"], d2["eff_range"], d2["Z"], d2["X"], d2["y"], d2["family"], d2["link"], stream=stream) as c2:
        c2.mcml_full(d2["start"], mcnr=False, m=256, maxiter=3, warmup=100, tol=0.0, lambda_=5.0, maxsteps=10, target_accept=0.9, seed=7, chains=256, maxfun=40)
d = synth.stepped_wedge(40, 8, 50)
with api.Context(d["cov"], d["data"], d["eff_range"], d["Z"], d["X"], d["y"], d["family"], d["link"], stream=stream) as ctx:
    kw = dict(mcnr=True, m=512, warmup=100, tol=0.0, lambda_=0.5, maxsteps=10, target_accept=0.9, seed=7, chains=512, maxfun=40)
    ctx.mcml_full(d["start"], maxiter=1, **kw)
    out = []
    for rep in range(8):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ctx.mcml_full(d["start"], maxiter=3, **kw)
        torch.cuda.synchronize(); out.append((time.perf_counter() - t0) / 3 * 1e3)
    print(mode, " ".join("%.1f" % v for v in out))
