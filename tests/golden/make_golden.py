"""Generates tests/golden/mcml_golden.json.

The reference (an R package with un-vendored C++ dependencies) cannot be built or
run in this image and ships no tests, fixtures or golden vectors of its own, so
these vectors are produced by the CPU oracle (oracle/mcml_oracle.c), which is
pinned separately against closed forms, scipy and libstdc++ (tests/test_oracle_*).
They freeze the oracle's outputs on small seeded inputs: a change in the oracle
or in the product that moves any of them is caught.  PARITY UNPINNED against the
real glmmrBase / rminqa (see oracle/mcml_oracle.h).

Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from glmmrmcml_amd import synth          # noqa: E402
from oracle import oracle as orc         # noqa: E402

CASES = {
    "crt_binomial": (synth.cluster_rct, dict(ncl=4, nt=3, nind=3, seed=11)),
    "crt_poisson": (synth.cluster_rct, dict(ncl=4, nt=3, nind=3, seed=12, family="poisson")),
    "geo_gaussian": (synth.geospatial, dict(n=20, seed=13)),
    "sw_binomial_ar1": (synth.stepped_wedge, dict(ncl=4, nt=3, nind=3, seed=14)),
}


def main():
    orc.build()
    out = {"logpdf": [], "cases": {}}
    rng = np.random.default_rng(2024)
    for fl in (1, 3, 7):
        for _ in range(6):
            y = float(rng.integers(0, 2)) if fl == 3 else (float(rng.integers(0, 9)) if fl == 1 else float(rng.normal()))
            mu, vp = float(rng.normal()), float(rng.uniform(0.5, 2.0))
            out["logpdf"].append([fl, y, mu, vp, orc.logpdf(y, mu, vp, fl)])
    for name, (gen, kw) in CASES.items():
        d = gen(**kw)
        fl = orc.flink(d["family"], d["link"])
        L = orc.gen_D(d["cov"], d["data"], d["eff_range"], d["theta"], chol=True)
        ZL = d["Z"] @ L
        xb = d["X"] @ d["beta"]
        r = np.random.default_rng(7)
        v = r.normal(size=d["Q"]) * 0.5
        u = np.asfortranarray(L @ r.normal(size=(d["Q"], 5)))
        s, flags, probs, dg = orc.hmc_chain(xb, ZL, d["y"], d["sigma"], fl, 6, 5, 0.4, 5, 0.9, 31337, chain_id=2,
                                            iter_idx=1, adapt=4)
        m = orc.mcnr(d["X"], d["Z"], d["y"], u, d["beta"], d["sigma"], d["family"], d["link"])
        out["cases"][name] = dict(
            gen=gen.__name__, kw=kw, v=v.tolist(), u=u.tolist(),
            mvn_ll=orc.mvn_ll(d["cov"], d["data"], d["eff_range"], d["theta"], u),
            log_prob=orc.log_prob(xb, ZL, d["y"], d["sigma"], fl, v),
            log_grad=orc.log_grad(xb, ZL, d["y"], d["sigma"], fl, v).tolist(),
            loglik=orc.model_loglik(d["Z"], xb, d["y"], u, d["sigma"], fl),
            hmc=dict(warmup=6, nsamp=5, lambda_=0.4, max_steps=5, target=0.9, seed=31337, chain=2, iter=1, adapt=4,
                     flags=[int(f) for f in flags], probs=probs.tolist(), last=(L @ s[:, -1]).tolist(), e=dg["e"]),
            mcnr=dict(beta=m["beta"].tolist(), sigma=m["sigma"]),
        )
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "mcml_golden.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=0)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
