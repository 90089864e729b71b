"""Per-leapfrog time of the sampler for few chains (chains = 1 is the reference's layout): the streamed products of
dgemm_skinny.h against the 128-column MFMA tiles (GLMMR_MCML_SKINNY=0).  usage: python scripts/time_fewchains.py [n]"""
import os, sys, time
sys.path.insert(0, ".")
import numpy as np
from glmmrmcml_amd import api, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
d = synth.geospatial(n, seed=1)
ctx = api.Context(d["cov"], d["data"], d["eff_range"], d["Z"], d["X"], d["y"], d["family"], d["link"])
ctx.update_L(d["theta"])
for C in (1, 4, 8, 12, 16, 17):
    for sk in ("1", "0"):
        os.environ["GLMMR_MCML_SKINNY"] = sk
        ctx.hmc_sample(d["beta"], d["sigma"], 5, C, 5.0, 10, 0.9, seed=1, chains=C)
        ctx.profile(enable=True, reset=True)
        t0 = time.time()
        dg = ctx.hmc_sample(d["beta"], d["sigma"], 60, C, 5.0, 10, 0.9, seed=1, chains=C)
        dt = time.time() - t0
        pr = ctx.profile(enable=False)
        print(f"chains {C:3d} skinny={sk}: {dt * 1e3:.1f} ms for 61 proposals; forward {pr['fwd_ms'] / max(pr['fwd_n'], 1) * 1e3:.1f} us "
              f"backward {pr['bwd_ms'] / max(pr['bwd_n'], 1) * 1e3:.1f} us per launch; accept {dg['accept_rate']:.3f}", flush=True)
