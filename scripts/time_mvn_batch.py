"""mvn_ll at k candidate thetas side by side (evaluation lanes, mvn.hip mvn_loglik_batch): ms per round and per evaluation.
usage: python scripts/time_mvn_batch.py [Q=5000] [m=1024] [k ...]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from glmmrmcml_amd import api, synth
Q = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
ks = [int(x) for x in sys.argv[3:]] or [1, 2, 4, 8, 16]
d = synth.geospatial(Q, seed=1)
ctx = api.Context(d["cov"], d["data"], d["eff_range"])
ctx.set_u(np.asfortranarray(np.random.default_rng(1).standard_normal((Q, m))))
th = lambda i: d["theta"] * (1 + 0.01 * (i % 17))
ref = [ctx.mvn_ll(th(i)) for i in range(max(ks))]
for _ in range(3): ctx.mvn_ll(th(1))
t0 = time.perf_counter(); n = 10
for i in range(n): ctx.mvn_ll(th(i))
base = (time.perf_counter() - t0) / n
print("Q=%d m=%d sequential: %.3f ms per evaluation" % (Q, m, base * 1e3), flush=True)
for k in ks:
    T = np.array([th(i) for i in range(k)])
    for _ in range(3): got = ctx.mvn_ll_batch(T)          # eager, capture, replay
    assert np.allclose(got, np.array(ref[:k]), rtol=1e-12, atol=0), (got, ref[:k])      # regrouped sums: equal to rounding
    t0 = time.perf_counter(); n = 6
    for i in range(n): ctx.mvn_ll_batch(T)
    dt = (time.perf_counter() - t0) / n
    print("  k=%2d: %.3f ms per round = %.3f ms per evaluation (x%.2f)" % (k, dt * 1e3, dt * 1e3 / k, base * k / dt), flush=True)
