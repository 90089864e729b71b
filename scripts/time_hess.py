"""mcml_hess (src/mcml_optim.cpp:263-285) on the bench workload: ms per call
usage: python scripts/time_hess.py [n=5000] [m=1024]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from glmmrmcml_amd import api, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
d = synth.geospatial(n, seed=1)
with api.Context(d["cov"], d["data"], d["eff_range"], d["Z"], d["X"], d["y"], d["family"], d["link"]) as ctx:
    ctx.set_u(np.asfortranarray(np.random.default_rng(1).standard_normal((n, m)) * 0.3))
    ctx.mcml_hess(d["start"], tol=1e-4)
    t0 = time.perf_counter(); H = ctx.mcml_hess(d["start"], tol=1e-4); dt = time.perf_counter() - t0
    print("mcml_hess n=%d m=%d: %.1f ms" % (n, m, dt * 1e3), np.round(np.diag(H), 3))
