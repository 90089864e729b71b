"""Caller-side bookkeeping of ModelMCML (glmmrmcml_amd/model.py, mirroring R/R6ModelExtMCML.R:103-845) with a
recording backend in place of the GPU exports: start-vector assembly, argument lists per option, index sets,
standard errors, cAIC / R-squared plumbing, checks on y.  No GPU."""
import numpy as np
import pytest

from glmmrmcml_amd import synth
from glmmrmcml_amd.model import ModelMCML, _dhdmu


class FakeCtx:
    def __init__(self, be, cov, data, eff):
        self.be = be

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def gen_D(self, theta, chol=False):
        Q = self.be.Q
        D = np.eye(Q) * float(np.sum(theta))
        return np.linalg.cholesky(D) if chol else D


class FakeBackend:
    """records every export call; returns shapes the real exports return"""

    def __init__(self, P, R, Q, m=7):
        self.P, self.R, self.Q, self.m = P, R, Q, m
        self.calls = []
        rng = np.random.default_rng(0)
        self.u = rng.normal(size=(Q, m))

    def Context(self, cov, data, eff):
        return FakeCtx(self, cov, data, eff)

    def _rec(self, name, args, kw):
        self.calls.append((name, args, kw))

    def mcml_full(self, *a, **k):
        self._rec("mcml_full", a, k)
        return dict(beta=np.full(self.P, 0.5), theta=np.full(self.R, 0.2), sigma=0.9, converged=True, u=self.u)

    def mcmc_sample(self, *a, **k):
        self._rec("mcmc_sample", a, k)
        return self.u

    def gen_u_samples(self, *a, **k):
        self._rec("gen_u_samples", a, k)
        return self.u

    def mcml_optim(self, *a, **k):
        self._rec("mcml_optim", a, k)
        return dict(beta=np.full(self.P, 0.5), theta=np.full(self.R, 0.2), sigma=0.9)

    def mcml_optim_sparse(self, *a, **k):
        self._rec("mcml_optim_sparse", a, k)
        return dict(beta=np.full(self.P, 0.5), theta=np.full(self.R, 0.2), sigma=0.9, Ap=np.zeros(self.Q + 1, dtype=np.int32),
                    Ai=np.zeros(0, dtype=np.int32), Ax=np.zeros(0), D=np.full(self.Q, 0.4))

    def mcml_simlik(self, *a, **k):
        self._rec("mcml_simlik", a, k)
        return dict(beta=np.full(self.P, 0.6), theta=np.full(self.R, 0.3), sigma=0.8)

    def mcml_hess(self, *a, **k):
        self._rec("mcml_hess", a, k)
        return np.eye(self.P + self.R) * 4.0

    def aic_mcml(self, *a, **k):
        self._rec("aic_mcml", a, k)
        return 123.0

    def mcml_la(self, *a, **k):
        self._rec("mcml_la", a, k)
        return dict(beta=np.full(self.P, 0.5), theta=np.full(self.R, 0.2), sigma=0.9, se=np.full(self.P + self.R + 1, 0.1),
                    u=np.zeros((self.Q, 1)))

    mcml_la_nr = mcml_la


def _model(family="binomial", **kw):
    d = synth.cluster_rct(ncl=4, nt=3, nind=3, seed=1, family=family) if family != "gaussian" else synth.geospatial(12, seed=2)
    be = FakeBackend(d["P"] if "P" in d else d["X"].shape[1], 2, d["Z"].shape[1])
    m = ModelMCML(d["cov"], d["data"], d["eff_range"], d["Z"], d["X"], d["family"], d["link"], d["beta"], d["theta"],
                  var_par=1.3, backend=be, **kw)
    return d, be, m


def test_start_vector_gets_a_trailing_one_for_binomial_and_poisson():
    d, be, m = _model("binomial")
    m.MCML(d["y"], verbose=False)
    name, a, k = be.calls[0]
    assert name == "mcml_full"
    start = a[8]
    assert start.size == m.X.shape[1] + 2 + 1 and start[-1] == 1.0           # R6ModelExtMCML.R:179
    assert np.allclose(start[:-1], np.r_[d["beta"], d["theta"]])
    assert k["mcnr"] is True and k["m"] == 250 and k["warmup"] == 500 and k["lambda_"] == 5.0
    assert k["maxsteps"] == 100 and k["target_accept"] == 0.95 and k["tol"] == 1e-2 and k["maxiter"] == 30
    with pytest.raises(ValueError, match="wrong number of starting values"):
        m.MCML(d["y"], start=np.zeros(3), verbose=False)


def test_gaussian_start_uses_var_par_and_reports_sigma():
    d, be, m = _model("gaussian")
    fit = m.MCML(d["y"], verbose=False, method="mcem")
    start = be.calls[0][1][8]
    assert start[-1] == 1.3 and start.size == 1 + 2 + 1                         # :163-166
    assert be.calls[0][2]["mcnr"] is False
    assert fit["coefficients"]["par"][:4] == ["b1", "cov1", "cov2", "sigma"]
    assert fit["theta"][-1] == 0.9                                              # sigma copied back (:413)
    # aic receives beta + sigma (mf_parInd) and the covariance parameters (:540-551)
    aic_call = [c for c in be.calls if c[0] == "aic_mcml"][0]
    assert np.allclose(aic_call[1][9], [0.5, 0.9]) and np.allclose(aic_call[1][10], [0.2, 0.2])
    with pytest.raises(ValueError):
        m.MCML(d["y"], start=np.zeros(3), verbose=False)


def test_y_checks():
    d, be, m = _model("binomial")
    with pytest.raises(ValueError, match="y must be 0 or 1"):
        m.MCML(np.full(m.n(), 2.0), verbose=False)
    d, be, m = _model("poisson")
    with pytest.raises(ValueError, match="integer"):
        m.MCML(np.full(m.n(), 0.5), verbose=False)
    with pytest.raises(ValueError, match="se.method"):
        m.MCML(d["y"], se_method="perm", verbose=False)


def test_stepwise_loop_calls_sampler_then_optim_and_rebuilds_L_from_previous_theta():
    d, be, m = _model("poisson")
    fit = m.MCML(d["y"], sampler="stepwise", verbose=False, max_iter=3, tol=1e-2)
    names = [c[0] for c in be.calls]
    assert names[:2] == ["mcmc_sample", "mcml_optim"]
    # the fake optimiser returns the same estimates every time: the loop stops after the second pass
    assert names.count("mcml_optim") == 2 and fit["converged"] and fit["iter"] == 2
    # first L = chol(D(start cov pars)) = sqrt(0.35) I; second L uses thetanew = the PREVIOUS theta (:317)
    L1 = be.calls[0][1][1]; L2 = be.calls[2][1][1]
    assert np.allclose(L1, np.eye(m.Z.shape[1]) * np.sqrt(0.35))
    assert np.allclose(L2, np.eye(m.Z.shape[1]) * np.sqrt(0.35))
    assert be.calls[1][2]["mcnr"] is True


def test_nuts_loop_is_the_usestan_branch_with_gen_u_samples():
    """usestan = TRUE (R6ModelExtMCML.R:234-257): data$Z = Z L, data$Xb from the current beta, data$sigma; the draws
    come back as L %*% t(gamma) and go to mcml_optim"""
    d, be, m = _model("gaussian")
    fit = m.MCML(d["y"], sampler="nuts", verbose=False, max_iter=3, tol=1e-2, seed=5)
    names = [c[0] for c in be.calls]
    assert names[:2] == ["gen_u_samples", "mcml_optim"] and "mcmc_sample" not in names
    a, k = be.calls[0][1], be.calls[0][2]
    assert np.array_equal(a[0], d["y"]) and a[1] is m.X and a[2] is m.Z           # (y, X, Z, L, beta, family, link)
    assert np.allclose(a[3], np.eye(m.Z.shape[1]) * np.sqrt(0.35))
    assert a[5] == "gaussian" and k["warmup_iter"] == m.mcmc_options["warmup"] and k["m"] == m.mcmc_options["samps"]
    assert k["sigma"] == pytest.approx(float(m._start(None)[0][-1])) and k["seed"] == 6
    assert fit["converged"]


def test_sparse_stepwise_uses_sparse_exports_and_ldl_factor():
    d, be, m = _model("poisson")
    m.MCML(d["y"], sampler="stepwise", sparse=True, verbose=False, max_iter=2, sim_lik_step=False)
    names = [c[0] for c in be.calls]
    assert "mcml_optim_sparse" in names and "mcml_optim" not in names
    call = [c for c in be.calls if c[0] == "mcml_optim_sparse"][0]
    Ap, Ai = call[1][3], call[1][4]
    assert Ap.dtype == np.int32 and Ap.size == m.Z.shape[1] + 1 and Ai.size == m.Z.shape[1]   # diagonal D
    # L = (I + strict lower) * sqrt(D) handed to the next sampler call (:313-315)
    L2 = [c for c in be.calls if c[0] == "mcmc_sample"][1][1][1]
    assert np.allclose(L2, np.eye(m.Z.shape[1]) * np.sqrt(0.4))


def test_simlik_step_and_hessian_se():
    d, be, m = _model("binomial")
    fit = m.MCML(d["y"], verbose=False, sim_lik_step=True, se_method="lik", options=dict(fd_tol=1e-3, trace=1))
    names = [c[0] for c in be.calls]
    assert names == ["mcml_full", "mcml_simlik", "mcml_hess", "aic_mcml"]
    assert "mcnr" not in be.calls[1][2]                                          # defect D7 not reproduced
    assert be.calls[2][2]["tol"] == 1e-3                                         # fd_tol forwarded (defect D12 fixed)
    P = m.X.shape[1]
    assert np.allclose(fit["theta"][:P], 0.6) and np.allclose(fit["theta"][P:P + 2], 0.3)
    assert fit["hessian"] is True and np.allclose(fit["coefficients"]["SE"][:P + 2], 0.5)   # sqrt(1/4)
    lo, up = fit["coefficients"]["lower"][0], fit["coefficients"]["upper"][0]
    assert up - lo == pytest.approx(2 * 1.959964 * 0.5, rel=1e-6)
    assert fit["aic"] == 123.0 and 0 < fit["Rsq"]["marg"] <= fit["Rsq"]["cond"] < 1
    assert "Markov Chain Newton-Raphson with simulated likelihood step" in str(fit)


def test_approx_se_is_gls_information():
    d, be, m = _model("binomial")
    fit = m.MCML(d["y"], verbose=False)
    P = m.X.shape[1]
    beta = np.full(P, 0.5)
    w = _dhdmu(m.X @ beta, "binomial", "logit")
    S = np.diag(w) + 0.4 * m.Z @ m.Z.T
    want = np.sqrt(np.diag(np.linalg.inv(m.X.T @ np.linalg.solve(S, m.X))))
    assert np.allclose(fit["coefficients"]["SE"][:P], want)
    assert np.all(np.isnan(fit["coefficients"]["SE"][P:P + 2]))                  # covariance SEs are NA with "approx"
    assert fit["re_samps"].shape == (m.Z.shape[1], 7)
    # d's: row means and sds of the samples
    assert np.allclose(fit["coefficients"]["est"][P + 2:], be.u.mean(axis=1))
    assert np.allclose(fit["coefficients"]["SE"][P + 2:], be.u.std(axis=1, ddof=1))


def test_LA_dispatch_and_table():
    d, be, m = _model("poisson")
    f1 = m.LA(d["y"])
    f2 = m.LA(d["y"], method="nr", use_hess=True)
    assert [c[0] for c in be.calls if c[0].startswith("mcml_la")] == ["mcml_la", "mcml_la"]   # fake aliases both
    k = be.calls[0][2]
    assert k["tol"] == 1e-2 and k["usehess"] is False and k["trace"] == 0
    assert f1["method"] == "nloptim" and f2["method"] == "nr" and f1["iter"] == 0 and f1["converged"]
    P = m.X.shape[1]
    assert np.allclose(f2["coefficients"]["SE"][:P + 2], 0.1)                     # resb$se with use.hess
    assert np.all(np.isnan(f1["coefficients"]["SE"][P:P + 2]))
    assert "Laplace Approximation" in str(f1)
    with pytest.raises(ValueError):
        m.LA(d["y"], method="bfgs")


# Printed coefficient tables of the reference's own README (README.md:56-64, 89-97, 144-147; the fits themselves are
# not reproducible here -- R's RNG simulated the data).  They pin the post-processing print.mcml applies to
# (estimate, SE): z = est / SE, two-sided normal p, est -+ qnorm(0.975) SE.  All figures are rounded to 2 decimals in
# the README, so the comparison allows the rounding of the inputs to propagate.
README_ROWS = [
    # name, Estimate, Std. Err., z value, p value, 2.5% CI, 97.5% CI
    ("int", 0.59, 0.23, 2.57, 0.01, 0.14, 1.04), ("t1", -1.51, 0.27, -5.50, 0.00, -2.04, -0.97),
    ("t2", -1.18, 0.26, -4.55, 0.00, -1.69, -0.67), ("t3", -1.23, 0.26, -4.70, 0.00, -1.74, -0.72),
    ("t4", 1.61, 0.30, 5.29, 0.00, 1.01, 2.20), ("t5", -2.57, 0.36, -7.21, 0.00, -3.26, -1.87),
    ("1 | gr(cl)", 0.62, 0.14, 4.47, 0.00, 0.35, 0.89),
    ("int", 0.91, 0.23, 3.90, 0, 0.45, 1.37), ("t4", 1.44, 0.31, 4.69, 0, 0.84, 2.04),
    ("1 | gr(cl) * ar1(t).1", 0.79, 0.11, 7.22, 0, 0.58, 1.01), ("1 | gr(cl) * ar1(t).2", 0.62, 0.12, 5.31, 0, 0.39, 0.85),
    ("(Intercept)", 0.86, 0.06, 13.91, 0, 0.74, 0.99), ("1 | fexp(x, y).1", 0.26, 0.02, 10.57, 0, 0.21, 0.31),
]


def test_print_table_formulas_against_the_reference_readme():
    from glmmrmcml_amd.model import McmlFit
    est = np.array([r[1] for r in README_ROWS]); se = np.array([r[2] for r in README_ROWS])
    fit = McmlFit(coefficients=dict(par=[r[0] for r in README_ROWS] + ["d1"], est=np.r_[est, 0.0], SE=np.r_[se, np.nan],
                                    lower=np.r_[est - 1.959963984540054 * se, np.nan],
                                    upper=np.r_[est + 1.959963984540054 * se, np.nan]),
                  re_samps=np.zeros((1, 3)), method="mcnr", sim_lik=False, family="binomial", link="logit", m=250,
                  tol=0.005, aic=422.04, Rsq=dict(cond=0.24, marg=0.21), converged=True)
    rows = fit.table()
    assert len(rows) == len(README_ROWS)
    for got, want in zip(rows, README_ROWS):
        e, s = want[1], want[2]
        # inputs are known to +-0.005: propagate that into each derived figure, plus the README's own rounding
        dz = abs(e) / s * (0.005 / abs(e) + 0.005 / s) + 0.006
        assert abs(got[3] - want[3]) <= dz, (want[0], got[3], want[3])
        assert abs(got[4] - want[4]) <= 0.006
        assert abs(got[5] - want[5]) <= 0.005 + 1.96 * 0.005 + 0.006
        assert abs(got[6] - want[6]) <= 0.005 + 1.96 * 0.005 + 0.006
    txt = str(fit)
    # the same header / footer lines the README prints (README.md:53-67)
    assert "Number of Monte Carlo simulations per iteration: 250 with tolerance 0.005" in txt
    assert "cAIC: 422.04" in txt and "Approximate R-squared: Conditional: 0.24  Marginal: 0.21" in txt
    # duplicated names get .1 / .2 suffixes as print.mcml does
    names = [r[0] for r in rows]
    assert names.count("int.1") == 1 and names.count("int.2") == 1 and names.count("t4.1") == 1


_FAMLINK = [("poisson", "log"), ("poisson", "identity"), ("binomial", "logit"), ("binomial", "log"),
            ("binomial", "identity"), ("binomial", "probit"), ("gaussian", "identity"), ("gaussian", "log"),
            ("gamma", "log"), ("gamma", "inverse"), ("gamma", "identity"), ("beta", "logit")]


@pytest.mark.parametrize("k", range(12))
def test_dhdmu_is_one_table(orc, k):
    """the caller's gen_dhdmu restatement equals the oracle's maths::dhdmu restatement (and so csrc/glm.h's, which the
    GPU parity tests compare with the oracle) for every family/link case"""
    fam, link = _FAMLINK[k]
    assert orc.flink(fam, link) == k + 1          # mcmlmodel.h:83-85: the map keys are lower-case "gamma..."
    eta = np.linspace(0.05, 0.9, 9) if link in ("identity", "inverse") else np.linspace(-1.2, 1.1, 9)
    assert np.allclose(_dhdmu(eta, fam, link), orc.dhdmu(eta, k + 1), rtol=1e-14, atol=0)
