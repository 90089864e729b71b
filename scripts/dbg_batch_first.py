import sys, time
sys.path.insert(0, ".")
import numpy as np
from glmmrmcml_amd import api, synth
Q, m = int(sys.argv[1]), int(sys.argv[2])
order = sys.argv[3:]            # e.g. 8 1 4 : batch sizes in the order they are first used
d = synth.geospatial(Q, seed=1)
ctx = api.Context(d["cov"], d["data"], d["eff_range"])
ctx.set_u(np.asfortranarray(np.random.default_rng(1).standard_normal((Q, m))))
th = lambda i: d["theta"] * (1 + 0.01 * (i % 17))
for k in [int(x) for x in order]:
    T = np.array([th(i) for i in range(k)])
    f = (lambda: ctx.mvn_ll_batch(T)) if k > 1 else (lambda: ctx.mvn_ll(th(1)))
    for _ in range(8): f()
    t0 = time.perf_counter(); n = 6
    for i in range(n): f()
    dt = (time.perf_counter() - t0) / n
    print("k=%d: %.3f ms per round = %.3f ms per evaluation" % (k, dt * 1e3, dt * 1e3 / k), flush=True)
