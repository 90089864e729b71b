import sys
sys.path.insert(0, '.')
import numpy as np
from glmmrmcml_amd import api, synth
n = int(sys.argv[1]); m = int(sys.argv[2])
d = synth.geospatial(n, seed=3)
rng = np.random.default_rng(1)
u = np.asfortranarray(rng.standard_normal((n, m)))
print("start", flush=True)
print(api.mvn_ll(d["cov"], d["data"], d["eff_range"], d["theta"], u), flush=True)
