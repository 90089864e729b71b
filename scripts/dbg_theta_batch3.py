import sys
sys.path.insert(0, ".")
import numpy as np
from glmmrmcml_amd import api, synth
d = synth.geospatial(150, seed=9)
u = np.asfortranarray(np.random.default_rng(1).standard_normal((150, 24)))
with api.Context(d["cov"], d["data"], d["eff_range"]) as ctx:
    ctx.set_u(u)
    T = np.array([[0.25, 0.1], [0.321, 0.1], [0.25, 0.1284], [0.1947, 0.1], [0.25, 0.0779], [0.321, 0.1284], [0.321, 0.0779], [0.25, 0.2]])
    for t in T:
        try: print(t, ctx.mvn_ll(t))
        except Exception as e: print(t, "ERR", e)
    print(ctx.mvn_ll_batch(T))
    print(ctx.mvn_ll_batch(T[:4]))
    print(ctx.mvn_ll_batch(T[4:]))
    print(ctx.mvn_ll_batch(T[[2, 0]]))
