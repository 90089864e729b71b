"""config 4 (stepped wedge 40 x 8 x 50, m = 512): ms per hmc_sample and per mcml_full iteration with the fused block kernel of the
factored operator (default) and with GLMMR_MCML_CM_LFUSE=0, alternating in one process"""
import os, sys, time
sys.path.insert(0, ".")
import numpy as np
import torch
from glmmrmcml_amd import api, synth
stream = torch.cuda.current_stream().cuda_stream
d = synth.stepped_wedge(40, 8, 50)
with api.Context(d["cov"], d["data"], d["eff_range"], d["Z"], d["X"], d["y"], d["family"], d["link"], stream=stream) as ctx:
    kw = dict(mcnr=True, m=512, warmup=100, tol=0.0, lambda_=0.5, maxsteps=10, target_accept=0.9, seed=7, chains=512, maxfun=40)
    ctx.update_L(d["theta"])
    ctx.mcml_full(d["start"], maxiter=1, **kw)
    for rnd in range(3):
        for fuse in ("1", "0"):
            os.environ["GLMMR_MCML_CM_LFUSE"] = fuse
            ctx.update_L(d["theta"])
            ts = []
            for rep in range(4):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                ctx.hmc_sample(d["beta"], 1.0, 100, 512, 0.5, 10, 0.9, seed=7, chains=512, iter_idx=rep)
                torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            ctx.mcml_full(d["start"], maxiter=3, **kw)
            torch.cuda.synchronize(); full = (time.perf_counter() - t0) / 3 * 1e3
            print("LFUSE=%s: hmc_sample %s ms; mcml_full %.1f ms per iteration" % (fuse, " ".join("%.1f" % t for t in ts), full), flush=True)
