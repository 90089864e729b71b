"""mcml_simlik (src/mcml_optim.cpp:90-117) on a geospatial model: sequential BOBYQA against the batch schedule
usage: python scripts/time_simlik.py [n=2000] [m=256]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from glmmrmcml_amd import api, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 256
d = synth.geospatial(n, seed=1)
with api.Context(d["cov"], d["data"], d["eff_range"], d["Z"], d["X"], d["y"], d["family"], d["link"]) as ctx:
    ctx.update_L(d["theta"])
    ctx.hmc_sample(d["beta"], d["sigma"], 60, m, 5.0, 10, 0.9, seed=3, chains=m)
    for tb in (1, 8):
        ctx.mcml_simlik(d["start"], theta_batch=tb, maxfun=12)
        t0 = time.perf_counter(); r = ctx.mcml_simlik(d["start"], theta_batch=tb); dt = time.perf_counter() - t0
        print("theta_batch=%d: %.1f ms  beta %s theta %s sigma %.6f" % (tb, dt * 1e3, np.round(r["beta"], 7), np.round(r["theta"], 7), r["sigma"]), flush=True)
