import sys, time
sys.path.insert(0, ".")
import numpy as np
from glmmrmcml_amd import api, synth
Q, m = 5000, 1024
d = synth.geospatial(Q, seed=1)
ctx = api.Context(d["cov"], d["data"], d["eff_range"])
ctx.set_u(np.asfortranarray(np.random.default_rng(1).standard_normal((Q, m))))
th = lambda i: d["theta"] * (1 + 0.01 * (i % 17))
def single(tag):
    for _ in range(3): ctx.mvn_ll(th(1))
    t0 = time.perf_counter(); n = 10
    for i in range(n): ctx.mvn_ll(th(i))
    print(tag, "single: %.3f ms per evaluation" % ((time.perf_counter() - t0) / n * 1e3), flush=True)
single("before any batch")
T = np.array([th(i) for i in range(8)])
for _ in range(4): ctx.mvn_ll_batch(T)
single("after batch of 8  ")
for _ in range(2): ctx.mvn_ll_batch(T[:2])
single("after batch of 2  ")
