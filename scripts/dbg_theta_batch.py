import sys
sys.path.insert(0, ".")
import numpy as np
from glmmrmcml_amd import api, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 150
m = int(sys.argv[2]) if len(sys.argv) > 2 else 24
d = synth.geospatial(n, seed=9)
args = (d["cov"], d["data"], d["eff_range"], d["Z"], d["X"], d["y"], d["family"], d["link"])
kw = dict(mcnr=True, maxiter=1, warmup=15, tol=1e-12, lambda_=0.3, maxsteps=6, target_accept=0.9, seed=4242, chains=m, m=m)
res = {}
with api.Context(*args) as ctx:
    for tb in (1, 2, 4, 8):
        ctx.theta_log(enable=True)
        r = ctx.mcml_full(d["start"], theta_batch=tb, **kw)
        log = ctx.theta_log(enable=False)
        u = ctx.get_u()
        ll = ctx.mvn_ll(r["theta"])
        res[tb] = (r["theta"], ll, log.shape[0], ctx.shard_stats()["theta_rounds"])
        print("theta_batch", tb, r["theta"], "ll %.12f" % ll, "evals", log.shape[0], "best logged %.12f" % log[:, 2].max(), log[np.argmax(log[:, 2]), :2], flush=True)
    # values of the batch evaluation against single ones on the last u
    T = np.array([d["theta"] * (1 + 0.03 * k) for k in range(8)])
    print("batch ", ctx.mvn_ll_batch(T))
    print("single", np.array([ctx.mvn_ll(t) for t in T]))
