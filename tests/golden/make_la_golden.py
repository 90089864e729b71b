"""Generates tests/golden/la_golden.json: outputs of the oracle's Laplace path (oracle/la.py) on small seeded
designs -- functor values at fixed points, one mcnr_b step, and the mcml_la / mcml_la_nr drivers.

The reference cannot run here (R, glmmrBase, rminqa absent; it ships no fixtures): these freeze the ORACLE's
answers (PARITY UNPINNED against the real package).  mcml_la's scipy optimisation over (beta, v) takes about a
minute, which is why its answer is stored instead of being recomputed inside the GPU tests.

Run from the repo root:  python tests/golden/make_la_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from glmmrmcml_amd import synth          # noqa: E402
from oracle import la as ola             # noqa: E402

CASES = {
    "crt_poisson": (synth.cluster_rct, dict(ncl=6, nt=3, nind=8, family="poisson")),
    "crt_binomial": (synth.cluster_rct, dict(ncl=8, nt=4, nind=10, family="binomial", seed=77)),
}


def model(d):
    return ola.LaModel(d["cov"], d["data"], d["eff_range"], d["Z"], d["X"], d["y"], d["family"], d["link"], d["start"])


def main():
    out = {}
    for name, (gen, kw) in CASES.items():
        d = gen(**kw)
        rng = np.random.default_rng(11)
        v = rng.normal(size=d["Q"]) * 0.3
        beta = d["beta"] + rng.normal(size=d["P"]) * 0.1
        theta = d["theta"] * (1 + 0.3 * rng.random(2))
        m = model(d); f0 = m.la_objective(np.r_[beta, v])
        m = model(d); m.v = v.copy(); m.update_W(False); f1 = m.la_cov_objective(theta)
        m = model(d); m.v = v.copy(); f2 = m.la_btheta_objective(np.r_[beta, theta])
        m = model(d); m.v = v.copy(); m.update_W(True); m.mcnr_b()
        args = (d["cov"], d["data"], d["eff_range"], d["Z"], d["X"], d["y"], d["family"], d["link"], d["start"])
        nr = ola.mcml_la_nr(*args, maxiter=6)
        rec = dict(gen=gen.__name__, kw=kw, v=v.tolist(), beta=beta.tolist(), theta=theta.tolist(),
                   f_bv=f0, f_cov=f1, f_btheta=f2,
                   mcnr_b=dict(v=m.v.tolist(), beta=m.beta.tolist(), sigma=m.sigma),
                   la_nr=dict(maxiter=6, beta=nr["beta"].tolist(), theta=nr["theta"].tolist(), sigma=nr["sigma"],
                              v=nr["v"].tolist(), u=nr["u"].tolist(), iters=nr["iters"]))
        if name == "crt_poisson":
            la = ola.mcml_la(*args, maxiter=3, usehess=True)
            rec["la"] = dict(maxiter=3, beta=la["beta"].tolist(), theta=la["theta"].tolist(), sigma=la["sigma"],
                             v=la["v"].tolist(), u=la["u"].tolist(), se=la["se"].tolist(), iters=la["iters"])
        out[name] = rec
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "la_golden.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
