import sys
sys.path.insert(0, ".")
import numpy as np
from glmmrmcml_amd import api, synth
d = synth.geospatial(150, seed=9)
args = (d["cov"], d["data"], d["eff_range"], d["Z"], d["X"], d["y"], d["family"], d["link"])
kw = dict(mcnr=True, maxiter=1, warmup=15, tol=1e-12, lambda_=0.3, maxsteps=6, target_accept=0.9, seed=4242, chains=24, m=24)
with api.Context(*args) as ctx:
    ctx.theta_log(enable=True)
    r = ctx.mcml_full(d["start"], theta_batch=4, **kw)
    log = ctx.theta_log(enable=False)
    print(np.round(log, 6))
