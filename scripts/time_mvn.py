"""times the theta-step objective (mvn_ll: covariance build + Cholesky + TRSM + reductions) alone.
usage: python scripts/time_mvn.py [n=5000] [m=1024] [reps=10]"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from glmmrmcml_amd import api, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
d = synth.geospatial(n)
ctx = api.Context(d["cov"], d["data"], d["eff_range"], d["Z"], d["X"], d["y"], d["family"], d["link"])
rng = np.random.default_rng(3)
ctx.set_u(np.asfortranarray(rng.standard_normal((n, m))))
v = ctx.mvn_ll(d["theta"])
ts = []
for r in range(reps):
    t0 = time.perf_counter(); v = ctx.mvn_ll(d["theta"] * (1.0 + 0.01 * r)); ts.append(time.perf_counter() - t0)
ts = np.array(ts) * 1e3
F = n ** 3 / 3.0 + float(n) ** 2 * m
print("mvn_ll n=%d m=%d: min %.3f ms  median %.3f ms  -> %.2f TFLOP/s algorithmic (Q^3/3 + Q^2 m = %.3g flop), value %.6f"
      % (n, m, ts.min(), np.median(ts), F / (np.median(ts) * 1e-3) / 1e12, F, v), flush=True)
