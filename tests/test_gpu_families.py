"""Every family / link case of the reference's switch (moremaths.h:26-102, mcmlmodel.h:169-276: flink 1..12; the
beta-logit score calls boost::math::digamma, restated as recurrence + asymptotic series identically in
oracle/mcml_oracle.c and csrc/glm.h) through the GPU path vs the oracle: log_prob,
log_grad, the Monte-Carlo log-likelihood, the MCNR statistics and one HMC chain's accept decisions.
Tolerance 1e-10 relative (f64, different summation order)."""
import numpy as np
import pytest

from glmmrmcml_amd import api, synth

pytestmark = pytest.mark.gpu


def _design(family, link, seed=5):
    """small cluster design whose linear predictor stays inside the link's domain"""
    d = synth.cluster_rct(ncl=6, nt=3, nind=6, seed=seed, family="poisson")
    rng = np.random.default_rng(seed + 100)
    n, P = d["n"], d["P"]
    X = d["X"]
    beta = np.zeros(P)
    theta = np.array([0.05, 0.03])                      # tiny random effects: eta stays near X beta
    centre = {"log": 0.3, "identity": 0.5, "logit": 0.2, "probit": 0.1, "inverse": 1.5}[link]
    if family == "binomial" and link == "log":
        centre = -1.0                                   # exp(eta) must stay below 1
    if family == "poisson" and link == "identity":
        centre = 3.0
    if family == "gamma" and link == "identity":
        centre = 2.0
    if family == "gaussian" and link == "log":
        centre = 1.5        # the reference logs y twice (mcmlmodel.h:90 and moremaths.h:81): keep log(y) > 0
    beta[1:] = centre                                   # the period columns partition the rows
    beta[0] = 0.05
    eta = X @ beta
    if family == "poisson":
        mu = np.exp(eta) if link == "log" else eta
        y = rng.poisson(mu).astype(float)
    elif family == "binomial":
        p = {"logit": 1 / (1 + np.exp(-eta)), "log": np.exp(eta), "identity": eta,
             "probit": 0.5 * (1 + np.vectorize(__import__("math").erf)(eta / np.sqrt(2)))}[link]
        y = (rng.random(n) < p).astype(float)
    elif family == "gaussian":
        y = eta + 0.3 * rng.normal(size=n) if link == "identity" else np.exp(eta + 0.1 * rng.normal(size=n))
    elif family == "gamma":
        mu = {"log": np.exp(eta), "inverse": 1 / eta, "identity": eta}[link]
        y = rng.gamma(shape=2.0, scale=mu / 2.0)
    else:                                               # beta
        mu = 1 / (1 + np.exp(-eta))
        y = np.clip(rng.beta(mu * 5, (1 - mu) * 5), 1e-3, 1 - 1e-3)
    return dict(d, family=family, link=link, y=y, beta=beta, theta=theta)


CASES = [("poisson", "log", 1.0), ("poisson", "identity", 1.0), ("binomial", "logit", 1.0), ("binomial", "log", 1.0),
         ("binomial", "identity", 1.0), ("binomial", "probit", 1.0), ("gaussian", "identity", 0.7),
         ("gaussian", "log", 0.6), ("gamma", "log", 2.0), ("gamma", "inverse", 2.0), ("gamma", "identity", 2.0),
         ("beta", "logit", 4.0)]


@pytest.mark.parametrize("family,link,vp", CASES)
def test_log_prob_grad_loglik_mcnr(orc, family, link, vp):
    d = _design(family, link)
    fl = orc.flink(family, link)
    assert fl == CASES.index((family, link, vp)) + 1
    Lo = orc.gen_D(d["cov"], d["data"], d["eff_range"], d["theta"], chol=True)
    ZL, xb = d["Z"] @ Lo, d["X"] @ d["beta"]
    yo = np.log(d["y"]) if fl == 8 else d["y"]          # mcmlmodel.h:89-91: the model keeps log(y)
    rng = np.random.default_rng(1)
    V = rng.normal(size=(d["Q"], 4)) * 0.5
    with api.Context(d["cov"], d["data"], d["eff_range"], d["Z"], d["X"], d["y"], family, link) as ctx:
        ctx.update_L(d["theta"])
        lp, G = ctx.log_prob_grad(d["beta"], vp, V)
        for c in range(4):
            lo = orc.log_prob(xb, ZL, yo, vp, fl, V[:, c])
            go = orc.log_grad(xb, ZL, yo, vp, fl, V[:, c])
            assert lp[c] == pytest.approx(lo, rel=1e-10)
            assert np.abs(G[:, c] - go).max() < 1e-10 * max(1.0, np.abs(go).max())
        u = np.asfortranarray(Lo @ V)
        ctx.set_u(u)
        assert ctx.loglik(d["beta"], vp) == pytest.approx(orc.model_loglik(d["Z"], xb, yo, u, vp, fl), rel=1e-10)
        r = ctx.mcnr(d["beta"], vp)
        ro = orc.mcnr(d["X"], d["Z"], yo, u, d["beta"], vp, family, link)
        assert np.allclose(r["beta"], ro["beta"], rtol=1e-8, atol=1e-10)
        assert r["sigma"] == pytest.approx(ro["sigma"], rel=1e-9)


@pytest.mark.parametrize("family,link,vp", [CASES[1], CASES[3], CASES[5], CASES[7], CASES[8], CASES[9], CASES[11]])
def test_hmc_chain_decisions(orc, family, link, vp):
    d = _design(family, link, seed=9)
    fl = orc.flink(family, link)
    Lo = orc.gen_D(d["cov"], d["data"], d["eff_range"], d["theta"], chol=True)
    ZL, xb = d["Z"] @ Lo, d["X"] @ d["beta"]
    yo = np.log(d["y"]) if fl == 8 else d["y"]
    warm, nsamp, lam, ms, ta, seed = 8, 6, 0.2, 5, 0.9, 4242
    with api.Context(d["cov"], d["data"], d["eff_range"], d["Z"], d["X"], d["y"], family, link) as ctx:
        ctx.update_L(d["theta"])
        diag, flags, probs = ctx.hmc_sample(d["beta"], vp, warm, nsamp, lam, ms, ta, seed, chains=2, adapt=6,
                                            want_trace=True)
        u = ctx.get_u()
    for c in range(2):
        so, fo, po, _ = orc.hmc_chain(xb, ZL, yo, vp, fl, warm, 3, lam, ms, ta, seed, chain_id=c, adapt=6)
        assert np.array_equal(flags[c], fo)
        assert np.abs(probs[c] - po).max() < 1e-9
        uo = Lo @ so[:, 1:]
        assert np.abs(u[:, c * 3:(c + 1) * 3] - uo).max() < 1e-8 * max(1.0, np.abs(uo).max())


def test_beta_family_logpdf(orc):
    """flink 12: the log-pdf (lgamma form, moremaths.h:95-99) at a second dispersion value"""
    d = _design("beta", "logit")
    fl = orc.flink("beta", "logit")
    assert fl == 12
    Lo = orc.gen_D(d["cov"], d["data"], d["eff_range"], d["theta"], chol=True)
    xb = d["X"] @ d["beta"]
    rng = np.random.default_rng(2)
    u = np.asfortranarray(Lo @ rng.normal(size=(d["Q"], 3)))
    with api.Context(d["cov"], d["data"], d["eff_range"], d["Z"], d["X"], d["y"], "beta", "logit") as ctx:
        ctx.set_u(u)
        assert ctx.loglik(d["beta"], 4.0) == pytest.approx(orc.model_loglik(d["Z"], xb, d["y"], u, 4.0, fl), rel=1e-10)
