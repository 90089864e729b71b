"""Oracle restatement of the Laplace path (oracle/la.py) checked against independent maths (CPU).

The reference has no tests for this path and its dependencies are not in the image (parity
unpinned); these known-answer checks pin the restatement itself:
  * gaussian/identity with Z = I: the joint mode of (beta, v) and logdet(ZL'WZL + I) have closed forms;
  * mcnr_b's v increment is a Newton step on the LA_likelihood surface in v when D0 = I (L = I);
  * the functors agree with a direct numpy evaluation.
"""
import numpy as np
import pytest

from oracle import la as ola
from oracle import oracle as orc
from glmmrmcml_amd import synth


def _small_geo(n=24, seed=5):
    return synth.geospatial(n, seed=seed)


def test_la_objective_matches_direct_numpy():
    d = _small_geo()
    start = np.r_[d["beta"], d["theta"], 0.7]
    m = ola.LaModel(d["cov"], d["data"], d["eff_range"], d["Z"], d["X"], d["y"], d["family"], d["link"], start)
    rng = np.random.default_rng(1)
    v = rng.normal(size=m.Q) * 0.3
    b = np.array([0.4])
    got = m.la_objective(np.r_[b, v])
    eta = d["X"] @ b + (d["Z"] @ m.L) @ v
    # moremaths.h:76 with its 3.141593 literal; var_par = 1
    ll = np.sum(-np.log(1.0) - 0.5 * np.log(2 * 3.141593) - 0.5 * (d["y"] - eta) ** 2)
    assert got == pytest.approx(-(ll - 0.5 * v @ v), rel=1e-12)


def test_la_cov_objective_logdet_closed_form():
    d = _small_geo()
    start = np.r_[d["beta"], d["theta"], 1.0]
    m = ola.LaModel(d["cov"], d["data"], d["eff_range"], d["Z"], d["X"], d["y"], d["family"], d["link"], start)
    rng = np.random.default_rng(2)
    m.v = rng.normal(size=m.Q) * 0.2
    th = np.array([0.3, 0.15]); sg = 0.8
    got = m.la_cov_objective(np.r_[th, sg])
    D = orc.gen_D(d["cov"], d["data"], d["eff_range"], th)
    L = np.linalg.cholesky(D)
    eta = m.xb + L @ m.v                                   # Z = I
    ll = np.sum(-np.log(sg) - 0.5 * np.log(2 * 3.141593) - 0.5 * ((d["y"] - eta) / sg) ** 2)
    # W was fixed by the constructor's update_W with var_par = 1: W = I (gaussian identity: dhdmu = 1)
    sign, ld = np.linalg.slogdet(D + np.eye(m.Q))          # det(L'L + I) = det(L L' + I)
    assert np.allclose(m.W, 1.0)
    assert got == pytest.approx(-(ll - 0.5 * m.v @ m.v - 0.5 * ld), rel=1e-10)


def test_mcnr_b_is_newton_step_for_poisson_identity_D():
    """gr-only covariance with theta = 1 gives L = D = I: then log_grad(v, usezl = false) IS the gradient of
    the LA_likelihood surface in v and (ZL'WZL + I) its negative Hessian: repeated mcnr_b steps (beta held)
    are Newton's method and must drive the gradient to round-off."""
    d = synth.cluster_rct(ncl=6, nt=3, nind=8, family="poisson")
    th = np.ones_like(d["theta"])
    start = np.r_[d["beta"], th]
    m = ola.LaModel(d["cov"], d["data"], d["eff_range"], d["Z"], d["X"], d["y"], d["family"], d["link"], start)
    assert np.allclose(m.L, np.eye(m.Q))

    def grad_v(v):
        return orc.log_grad(m.xb, m.ZL, m.y, m.var_par, m.fl, v)
    m.v = np.full(m.Q, 0.05)
    g0 = np.linalg.norm(grad_v(m.v))
    beta0 = m.beta.copy()
    gs = []
    for _ in range(8):
        m.xb = m.X @ beta0                                 # hold beta: look at the v block only
        m.update_W(True)
        m.mcnr_b()
        gs.append(np.linalg.norm(grad_v(m.v)))
    assert gs[-1] < 1e-9 * g0
    assert gs[5] < 1e-3 * gs[3]                            # quadratic tail


def test_drivers_reach_a_stationary_point_poisson():
    """end-to-end mcml_la_nr on a small cluster design: the returned (beta, theta) is a minimum of the final
    joint functor (LA_likelihood_btheta) the driver polishes with"""
    d = synth.cluster_rct(ncl=6, nt=3, nind=8, family="poisson")
    a = ola.mcml_la_nr(d["cov"], d["data"], d["eff_range"], d["Z"], d["X"], d["y"], d["family"], d["link"],
                       d["start"], maxiter=4)
    assert np.all(np.isfinite(a["beta"])) and np.all(a["theta"] >= 1e-6)
    assert a["u"].shape == (d["Q"],)
    m = ola.LaModel(d["cov"], d["data"], d["eff_range"], d["Z"], d["X"], d["y"], d["family"], d["link"], d["start"])
    m.v = a["v"].copy()
    x = np.r_[a["beta"], a["theta"]]
    f0 = m.la_btheta_objective(x)
    rng = np.random.default_rng(0)
    for _ in range(10):
        dx = rng.normal(size=x.size) * 1e-3
        xx = x + dx
        xx[m.P:] = np.maximum(xx[m.P:], 1e-6)
        assert m.la_btheta_objective(xx) >= f0 - 1e-9 * abs(f0)


def test_oracle_reproduces_la_golden():
    """the committed Laplace fixture (tests/golden/la_golden.json) is what the oracle computes today; the
    minute-long mcml_la optimisation is only checked for being a minimum of its final functor"""
    import json, os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "la_golden.json")) as f:
        G = json.load(f)
    for name, g in G.items():
        d = getattr(synth, g["gen"])(**g["kw"])
        mk = lambda: ola.LaModel(d["cov"], d["data"], d["eff_range"], d["Z"], d["X"], d["y"], d["family"], d["link"],
                                 d["start"])
        v = np.array(g["v"]); beta = np.array(g["beta"]); theta = np.array(g["theta"])
        assert mk().la_objective(np.r_[beta, v]) == pytest.approx(g["f_bv"], rel=1e-12)
        m = mk(); m.v = v.copy(); m.update_W(False)
        assert m.la_cov_objective(theta) == pytest.approx(g["f_cov"], rel=1e-12)
        m = mk(); m.v = v.copy()
        assert m.la_btheta_objective(np.r_[beta, theta]) == pytest.approx(g["f_btheta"], rel=1e-12)
        m = mk(); m.v = v.copy(); m.update_W(True); m.mcnr_b()
        assert np.allclose(m.v, g["mcnr_b"]["v"], rtol=1e-12, atol=1e-14)
        assert np.allclose(m.beta, g["mcnr_b"]["beta"], rtol=1e-12, atol=1e-14)
        nr = ola.mcml_la_nr(d["cov"], d["data"], d["eff_range"], d["Z"], d["X"], d["y"], d["family"], d["link"],
                            d["start"], maxiter=g["la_nr"]["maxiter"])
        assert np.allclose(nr["beta"], g["la_nr"]["beta"], atol=1e-6)
        assert np.allclose(nr["theta"], g["la_nr"]["theta"], atol=1e-6)
        if "la" in g:
            m = mk(); m.v = np.array(g["la"]["v"])
            x = np.r_[g["la"]["beta"], g["la"]["theta"]]
            f0 = m.la_btheta_objective(x)
            rng = np.random.default_rng(1)
            for _ in range(6):
                xx = x + rng.normal(size=x.size) * 1e-3
                xx[m.P:] = np.maximum(xx[m.P:], 1e-6)
                assert m.la_btheta_objective(xx) >= f0 - 1e-9 * abs(f0)
