import sys, time
sys.path.insert(0, ".")
import numpy as np
import torch
from glmmrmcml_amd import api, synth
stream = torch.cuda.current_stream().cuda_stream
what = sys.argv[1:]
if "main" in what:
    d3 = synth.geospatial(5000, seed=20240601)
    with api.Context(d3["cov"], d3["data"], d3["eff_range"], d3["Z"], d3["X"], d3["y"], d3["family"], d3["link"], stream=stream) as c3:
        c3.mcml_full(d3["start"], mcnr=True, m=1024, maxiter=2, warmup=100, tol=0.0, lambda_=5.0, maxsteps=10, target_accept=0.9, seed=1, chains=1024, maxfun=40)
        if "ll" in what: c3.mvn_ll(d3["theta"])
if "cfg2" in what:
    d2 = synth.geospatial(2000, seed=1)
    with api.Context(d2["cov"], d2["data"], d2["eff_range"], d2["Z"], d2["X"], d2["y"], d2["family"], d2["link"], stream=stream) as c2:
        c2.mcml_full(d2["start"], mcnr=False, m=256, maxiter=3, warmup=100, tol=0.0, lambda_=5.0, maxsteps=10, target_accept=0.9, seed=7, chains=256, maxfun=40)
d = synth.stepped_wedge(40, 8, 50)
with api.Context(d["cov"], d["data"], d["eff_range"], d["Z"], d["X"], d["y"], d["family"], d["link"], stream=stream) as ctx:
    kw = dict(mcnr=True, m=512, warmup=100, tol=0.0, lambda_=0.5, maxsteps=10, target_accept=0.9, seed=7, chains=512, maxfun=40)
    ctx.mcml_full(d["start"], maxiter=1, **kw)
    out = []
    for rep in range(6):
        if "prof" in what: ctx.profile(enable=True, reset=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ctx.mcml_full(d["start"], maxiter=3, **kw)
        torch.cuda.synchronize(); out.append((time.perf_counter() - t0) / 3 * 1e3)
        if "prof" in what: ctx.profile(enable=False)
    print(" ".join(what), ":", " ".join("%.1f" % v for v in out))
