import sys
sys.path.insert(0, '.')
import numpy as np
from glmmrmcml_amd import api, synth
for n, m in ((2048, 64), (2056, 64), (2500, 1024), (5000, 64), (5000, 1024)):
    d = synth.geospatial(n, seed=3)
    rng = np.random.default_rng(1)
    u = np.asfortranarray(rng.standard_normal((n, m)))
    with api.Context(d["cov"], d["data"], d["eff_range"]) as ctx:
        ctx.set_u(u)
        v = [ctx.mvn_ll(d["theta"]) for _ in range(3)]
    print(n, m, ["%.6f" % x for x in v], flush=True)
