"""Times the No-U-Turn sampler (csrc/nuts.h) on the bench workload (geospatial n = Q, C chains) or config 4.
usage: python scripts/time_nuts.py [n|cfg4] [chains] [warmup]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from glmmrmcml_amd import api, synth

which = sys.argv[1] if len(sys.argv) > 1 else "5000"
C = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
warm = int(sys.argv[3]) if len(sys.argv) > 3 else 20
d = synth.stepped_wedge(40, 8, 50) if which == "cfg4" else synth.geospatial(int(which), seed=1)
vp = d.get("sigma", 1.0)
ctx = api.Context(d["cov"], d["data"], d["eff_range"], d["Z"], d["X"], d["y"], d["family"], d["link"])
ctx.update_L(d["theta"])
ctx.nuts_sample(d["beta"], vp, 2, C, seed=3, chains=C)           # warm the allocations
ctx.profile(enable=True, reset=True)
t0 = time.time()
dg, tr = ctx.nuts_sample(d["beta"], vp, warm, C, seed=4, chains=C, want_trace=True)
dt = time.time() - t0
pr = ctx.profile(enable=False)
nl = dg["batched_leapfrogs"] + dg["stepsize_search_leapfrogs"]
gemm_ms = pr["fwd_ms"] + pr["bwd_ms"]
print(f"{which} n {d['n']} Q {d['Q']} chains {C} warmup {warm}: {dt:.3f} s, batched leapfrogs {nl} ({dt / nl * 1e3:.3f} ms each), "
      f"products {gemm_ms:.1f} ms = {gemm_ms / (dt * 1e3):.2f} of the wall time")
print("depth mean %.2f max %d ; leapfrogs per transition per chain mean %.1f ; chain-leapfrogs/s %.3e ; eps %.4f ; divergent %d"
      % (tr["depth"].mean(), tr["depth"].max(), tr["nleap"].mean(), tr["nleap"].sum() / dt, dg["mean_e"], dg["divergent"]))
t0 = time.time()
dg2 = ctx.hmc_sample(d["beta"], vp, warm, C, 1.0, 10, 0.9, seed=4, chains=C)
dt2 = time.time() - t0
print(f"fixed-length HMC, same warm-up: {dt2:.3f} s, {dg2['leapfrog_total'] / dt2:.3e} chain-leapfrogs/s")
