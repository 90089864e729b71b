import sys, time
sys.path.insert(0, '.')
import numpy as np
from glmmrmcml_amd import api, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
C = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
t0 = time.time(); d = synth.geospatial(n); print("synth", time.time() - t0, flush=True)
ctx = api.Context(d["cov"], d["data"], d["eff_range"], d["Z"], d["X"], d["y"], d["family"], d["link"])
t0 = time.time(); ctx.update_L(d["theta"]); print("update_L", time.time() - t0, flush=True)
t0 = time.time(); ctx.update_L(d["theta"]); print("update_L again", time.time() - t0, flush=True)
for warm in (2, 10):
    t0 = time.time()
    dg = ctx.hmc_sample(d["beta"], d["sigma"], warm, C, 5.0, 10, 0.9, seed=1, chains=C)
    dt = time.time() - t0
    lf = dg["leapfrog_total"]
    print(f"hmc warm={warm} chains={C}: {dt:.3f}s  leapfrog steps {lf}  {lf/dt:.3e} chain-steps/s  "
          f"{4.0*n*n*lf/dt/1e12:.2f} TFLOP/s algorithmic  acc={dg['accept_rate']:.3f} e={dg['mean_e']:.4f}", flush=True)
u = ctx.get_u()
t0 = time.time(); v = ctx.mvn_ll(d["theta"]); print("mvn_ll", time.time() - t0, v, flush=True)
t0 = time.time(); v = ctx.mvn_ll(d["theta"] * 1.1); print("mvn_ll again", time.time() - t0, v, flush=True)
t0 = time.time(); r = ctx.mcnr(d["beta"], d["sigma"]); print("mcnr", time.time() - t0, r["beta"], r["sigma"], flush=True)
t0 = time.time(); r = ctx.loglik(d["beta"], d["sigma"]); print("loglik", time.time() - t0, r, flush=True)
