"""single evaluations whose graph is captured again and again (the sample count changes): every re-capture must end up on a fast executable"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from glmmrmcml_amd import api, synth
Q = 5000
d = synth.geospatial(Q, seed=1)
ctx = api.Context(d["cov"], d["data"], d["eff_range"])
th = lambda i: d["theta"] * (1 + 0.01 * (i % 17))
for m in (1024, 1000, 1024, 960, 1024, 992):
    ctx.set_u(np.asfortranarray(np.random.default_rng(1).standard_normal((Q, m))))
    for i in range(12): ctx.mvn_ll(th(i))
    t0 = time.perf_counter(); n = 8
    for i in range(n): ctx.mvn_ll(th(i))
    print("m=%4d single: %.3f ms per evaluation" % (m, (time.perf_counter() - t0) / n * 1e3), flush=True)
