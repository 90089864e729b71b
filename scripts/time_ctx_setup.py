"""what the boundary's host buffers cost at config 3's size (n = Q = 5000, m = 1024): context creation (Z, X, y uploaded through
the library's pinned staging buffer, common.hip copy_h2d) and the download of u after a fit (Q x m doubles)"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import torch
from glmmrmcml_amd import api, synth
stream = torch.cuda.current_stream().cuda_stream
d = synth.geospatial(5000, seed=20240601)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ctx = api.Context(d["cov"], d["data"], d["eff_range"], d["Z"], d["X"], d["y"], d["family"], d["link"], stream=stream)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    ctx.update_L(d["theta"])
    torch.cuda.synchronize(); t2 = time.perf_counter()
    if rep == 2:
        ctx.hmc_sample(d["beta"], 1.0, 2, 1024, 5.0, 10, 0.9, seed=1, chains=1024)
        torch.cuda.synchronize(); t3 = time.perf_counter()
        u = ctx.get_u()
        torch.cuda.synchronize(); t4 = time.perf_counter()
        print("get_u (%d x %d doubles = %.0f MB): %.1f ms" % (u.shape[0], u.shape[1], u.nbytes / 1e6, (t4 - t3) * 1e3))
    print("context creation (Z %d x %d = %.0f MB uploaded): %.1f ms; first update_L %.1f ms" %
          (d["Z"].shape[0], d["Z"].shape[1], d["Z"].nbytes / 1e6, (t1 - t0) * 1e3, (t2 - t1) * 1e3))
    ctx.close()
