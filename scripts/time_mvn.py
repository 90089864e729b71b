import sys, time
sys.path.insert(0, '.')
import numpy as np
from glmmrmcml_amd import api, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
d = synth.geospatial(n)
rng = np.random.default_rng(0)
u = np.asfortranarray(rng.normal(size=(n, m)))
ctx = api.Context(d["cov"], d["data"], d["eff_range"])
ctx.set_u(u)
ctx.mvn_ll(d["theta"])
t0 = time.time()
for i in range(10):
    v = ctx.mvn_ll(d["theta"] * (1 + 0.01 * i))
print("mvn_ll avg ms", (time.time() - t0) / 10 * 1e3, v, flush=True)
