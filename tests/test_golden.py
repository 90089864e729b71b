"""Golden vectors (tests/golden/mcml_golden.json, made by tests/golden/make_golden.py from
the CPU oracle; the reference ships none): the oracle must still reproduce them bit for bit
on the CPU, and the HIP path must match them on the GPU (1e-10 relative; accept/reject exact)."""
import json
import os

import numpy as np
import pytest

from glmmrmcml_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))
G = json.load(open(os.path.join(HERE, "golden", "mcml_golden.json")))
GEN = dict(cluster_rct=synth.cluster_rct, geospatial=synth.geospatial, stepped_wedge=synth.stepped_wedge)


def _case(name):
    c = G["cases"][name]
    return c, GEN[c["gen"]](**c["kw"])


def test_oracle_reproduces_logpdf_vectors(orc):
    for fl, y, mu, vp, want in G["logpdf"]:
        assert orc.logpdf(y, mu, vp, int(fl)) == want


@pytest.mark.parametrize("name", sorted(G["cases"]))
def test_oracle_reproduces_golden(orc, name):
    c, d = _case(name)
    fl = orc.flink(d["family"], d["link"])
    L = orc.gen_D(d["cov"], d["data"], d["eff_range"], d["theta"], chol=True)
    ZL = d["Z"] @ L; xb = d["X"] @ d["beta"]
    u = np.asfortranarray(c["u"]); v = np.array(c["v"])
    assert orc.mvn_ll(d["cov"], d["data"], d["eff_range"], d["theta"], u) == c["mvn_ll"]
    assert orc.log_prob(xb, ZL, d["y"], d["sigma"], fl, v) == c["log_prob"]
    assert np.array_equal(orc.log_grad(xb, ZL, d["y"], d["sigma"], fl, v), np.array(c["log_grad"]))
    h = c["hmc"]
    s, flags, probs, dg = orc.hmc_chain(xb, ZL, d["y"], d["sigma"], fl, h["warmup"], h["nsamp"], h["lambda_"],
                                        h["max_steps"], h["target"], h["seed"], chain_id=h["chain"],
                                        iter_idx=h["iter"], adapt=h["adapt"])
    assert [int(f) for f in flags] == h["flags"] and np.array_equal(probs, np.array(h["probs"]))
    assert np.array_equal(L @ s[:, -1], np.array(h["last"]))


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(G["cases"]))
def test_hip_path_matches_golden(name):
    from glmmrmcml_amd import api
    c, d = _case(name)
    u = np.asfortranarray(c["u"]); v = np.array(c["v"])
    with api.Context(d["cov"], d["data"], d["eff_range"], d["Z"], d["X"], d["y"], d["family"], d["link"]) as ctx:
        ctx.update_L(d["theta"])
        ctx.set_u(u)
        assert abs(ctx.mvn_ll(d["theta"]) - c["mvn_ll"]) < 1e-10 * abs(c["mvn_ll"])
        assert abs(ctx.loglik(d["beta"], d["sigma"]) - c["loglik"]) < 1e-10 * abs(c["loglik"])
        m = ctx.mcnr(d["beta"], d["sigma"])
        assert np.allclose(m["beta"], c["mcnr"]["beta"], rtol=1e-9, atol=1e-11)
        assert abs(m["sigma"] - c["mcnr"]["sigma"]) < 1e-11 * c["mcnr"]["sigma"]
        lp, G_ = ctx.log_prob_grad(d["beta"], d["sigma"], v.reshape(-1, 1))
        assert abs(lp[0] - c["log_prob"]) < 1e-11 * abs(c["log_prob"])
        assert np.abs(G_[:, 0] - np.array(c["log_grad"])).max() < 1e-11 * max(1.0, np.abs(c["log_grad"]).max())
        h = c["hmc"]
        # chains = 3 so that local chain 2 carries the golden chain's global id 2
        diag, flags, probs = ctx.hmc_sample(d["beta"], d["sigma"], h["warmup"], 3 * h["nsamp"], h["lambda_"],
                                            h["max_steps"], h["target"], h["seed"], chains=3, iter_idx=h["iter"],
                                            adapt=h["adapt"], want_trace=True)
        assert [int(f) for f in flags[2]] == h["flags"]
        assert np.abs(probs[2] - np.array(h["probs"])).max() < 1e-9
        uu = ctx.get_u()
        assert np.abs(uu[:, 3 * h["nsamp"] - 1] - np.array(h["last"])).max() < 1e-8
