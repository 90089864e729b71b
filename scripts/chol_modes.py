"""mvn_ll of one dense block against numpy, for the factorisation variants (GLMMR_MCML_CHOL = default | nola | rec)"""
import sys
sys.path.insert(0, '.')
import numpy as np
from glmmrmcml_amd import api, synth
for n, m in ((700, 10), (1300, 37), (2049, 130)):
    d = synth.geospatial(n, seed=3)
    rng = np.random.default_rng(1)
    xy = np.c_[d["data"][:n], d["data"][n:]]
    D = synth._fexp_D(xy, d["theta"])
    L = np.linalg.cholesky(D)
    u = np.asfortranarray(L @ rng.standard_normal((n, m)))
    z = np.linalg.solve(L, u)
    want = np.mean(-0.5 * n * np.log(2 * np.pi) - np.log(np.diag(L)).sum() - 0.5 * (z ** 2).sum(0))
    got = api.mvn_ll(d["cov"], d["data"], d["eff_range"], d["theta"], u)
    print("n=%d m=%d  got %.10f  want %.10f  rel err %.2e" % (n, m, got, want, abs(got - want) / abs(want)), flush=True)
