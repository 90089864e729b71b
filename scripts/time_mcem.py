"""Times whole MCML iterations (mcml_full: sample + beta-step + theta-step + refresh) of the BASELINE configs on one GPU:
config 2 = gaussian geospatial n = 2000, m = 256; cfg4 = binomial stepped-wedge n = 16000, m = 512; cfg5 = poisson
longitudinal n = 20000, m = 1024 -- MCEM and MCNR.
usage: python scripts/time_mcem.py [n=2000|cfg4|cfg5] [m=256] [iters=3]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from glmmrmcml_amd import api, synth

which = sys.argv[1] if len(sys.argv) > 1 else "2000"
m = int(sys.argv[2]) if len(sys.argv) > 2 else 256
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 3
d = synth.stepped_wedge(40, 8, 50) if which == "cfg4" else synth.longitudinal(2000, 10) if which == "cfg5" else synth.geospatial(int(which), seed=1)
n = d["n"]
lam = 5.0 if which not in ("cfg4", "cfg5") else 0.5
ctx = api.Context(d["cov"], d["data"], d["eff_range"], d["Z"], d["X"], d["y"], d["family"], d["link"])
for mcnr in (False, True):
    kw = dict(mcnr=mcnr, m=m, maxiter=1, warmup=100, tol=0.0, verbose=False, lambda_=lam, maxsteps=10, target_accept=0.9,
              seed=7, chains=m, maxfun=40)
    ctx.mcml_full(d["start"], **kw)
    kw["maxiter"] = iters
    t0 = time.time()
    r = ctx.mcml_full(d["start"], **kw)
    dt = time.time() - t0
    print(f"{which}: n={n} m={m} {'MCNR' if mcnr else 'MCEM'}: {dt / iters * 1e3:.1f} ms per iteration = {m * iters / dt:.0f} simlik evals/s ({iters} iterations), "
          f"beta {np.round(r['beta'], 4)} theta {np.round(r['theta'], 4)}", flush=True)
