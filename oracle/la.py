"""Oracle restatement of the Laplace-approximation path (TEST INFRASTRUCTURE ONLY).

Follows, literally:
  functors  LA_likelihood / LA_likelihood_cov / LA_likelihood_btheta   likelihood.h:112-230
  steps     mcmloptim::la_optim / la_optim_cov / la_optim_bcov / hess_la / mcnr_b
                                                                        mcmloptim.h:116-195,238-293
  drivers   mcml_la / mcml_la_nr                                        src/mcml_la.cpp:28-290
  state     mcmlModel ctor, update_W(i, useL), log_grad(v, usezl=false) mcmlmodel.h:51-134,156-168

Reference behaviour kept on purpose (each is what the C++ does, not what one would write):
  * v (= u column 0) is the WHITENED random effect, yet update_W(useL=false) forms Z*v and
    log_grad(v, false) forms xb + Z*v and -D*v (mcmlmodel.h:121,165-166);
  * D_ is computed once in the constructor from the starting theta and never refreshed
    (update_D is commented out, src/mcml_la.cpp:247), so mcnr_b keeps using D(theta_start);
  * var_par starts at 1 whatever `start` holds (src/mcml_la.cpp:45,191);
  * the final joint fit appends sigma for the gaussian family only (mcmloptim.h:166-169).
PARITY UNPINNED: glmmrBase / rminqa are not in the image; the optimiser here is scipy's,
converged tightly on the same functors (see oracle/drivers.py).
"""
import numpy as np

from . import oracle as orc
from .drivers import _minimise, optimhess


def _has_var_par(fl):          # family_ == "gaussian" || "Gamma" || "beta"; the map keys make "Gamma" unreachable
    return fl in (7, 8, 12)


def _is_gaussian(fl):
    return fl in (7, 8)


def _vec(f, eta, code):
    return np.array([f(float(e), int(code)) for e in eta])


class LaModel:
    """mcmlModel + mcmloptim state of the Laplace drivers"""

    def __init__(self, cov, data, eff_range, Z, X, y, family, link, start):
        self.cov, self.data, self.eff = cov, data, eff_range
        self.Z = np.asarray(Z, float); self.X = np.asarray(X, float)
        self.family, self.link = family, link
        self.fl = orc.flink(family, link)
        self.lc = orc.link_code(link)
        self.y = np.log(np.asarray(y, float)) if self.fl == 8 else np.asarray(y, float)   # mcmlmodel.h:89-91
        self.n, self.P = self.X.shape
        self.Q = self.Z.shape[1]
        self.R = orc.cov_npar(cov)
        start = np.asarray(start, float)
        self.beta = start[:self.P].copy()
        self.theta = start[self.P:self.P + self.R].copy()
        self.sigma = start[self.P + self.R] if _is_gaussian(self.fl) else 0.0      # mcmloptim.h:30
        self.var_par = 1.0
        self.v = np.zeros(self.Q)
        self.L = orc.gen_D(cov, data, eff_range, self.theta, chol=True)
        self.ZL = self.Z @ self.L
        self.D0 = self.L @ self.L.T                                              # mcmlmodel.h:71, never refreshed
        self.xb = self.X @ self.beta
        self.W = np.ones(self.n)
        self.update_W(False)                                                     # mcmlmodel.h:94

    # mcmlmodel.h:120-134
    def update_W(self, useL):
        zu = (self.ZL if useL else self.Z) @ self.v
        w = _vec(orc.lib().orc_dhdmu, self.xb + zu, self.fl)
        nvar = 1.0
        if _is_gaussian(self.fl):
            nvar = self.var_par * self.var_par
        elif self.fl == 12:
            nvar = 1 + self.var_par
        self.W = 1.0 / (w * nvar)

    def _ll(self, ZL, v):
        eta = self.xb + ZL @ v
        return sum(orc.logpdf(self.y[i], eta[i], self.var_par, self.fl) for i in range(self.n))

    def _logdet_term(self, ZL):
        M = ZL.T @ (self.W[:, None] * ZL) + np.eye(self.Q)
        return 2.0 * np.sum(np.log(np.diag(np.linalg.cholesky(M))))

    # LA_likelihood (likelihood.h:112-140): par = (beta, v)
    def la_objective(self, par):
        par = np.asarray(par, float)
        v = par[self.P:]
        self.xb = self.X @ par[:self.P]
        self.v = v.copy()
        return -1.0 * (self._ll(self.ZL, v) - 0.5 * float(v @ v))

    # LA_likelihood_cov (likelihood.h:142-183): par = (theta[, var_par])
    def la_cov_objective(self, par):
        par = np.asarray(par, float)
        R = par.size - 1 if _has_var_par(self.fl) else par.size
        if _has_var_par(self.fl):
            self.var_par = par[R]
        try:
            L = orc.gen_D(self.cov, self.data, self.eff, par[:R], chol=True)
        except RuntimeError:
            return np.inf
        ZL = self.Z @ L
        return -1.0 * (self._ll(ZL, self.v) - 0.5 * float(self.v @ self.v) - 0.5 * self._logdet_term(ZL))

    # LA_likelihood_btheta (likelihood.h:185-230): par = (beta, theta[, var_par if gaussian])
    def la_btheta_objective(self, par):
        par = np.asarray(par, float)
        R = par.size - self.P - (1 if _is_gaussian(self.fl) else 0)
        if _is_gaussian(self.fl):
            self.var_par = par[-1]
        self.xb = self.X @ par[:self.P]
        self.update_W(False)
        try:
            L = orc.gen_D(self.cov, self.data, self.eff, par[self.P:self.P + R], chol=True)
        except RuntimeError:
            return np.inf
        ZL = self.Z @ L
        return -1.0 * (self._ll(ZL, self.v) - 0.5 * float(self.v @ self.v) - 0.5 * self._logdet_term(ZL))

    # ---- mcmloptim steps ----
    def la_optim(self):
        x0 = np.r_[self.beta, self.v]
        x, _ = _minimise(self.la_objective, x0, np.full(x0.size, -np.inf))
        self.la_objective(x)                       # leave the model at the optimum, as the last call would
        self.beta = x[:self.P].copy(); self.v = x[self.P:].copy()

    def la_optim_cov(self):
        x0 = self.theta.copy(); lo = np.full(self.R, 1e-6)
        if _has_var_par(self.fl):
            x0 = np.r_[x0, self.sigma]; lo = np.r_[lo, 0.0]
        x, _ = _minimise(self.la_cov_objective, x0, lo)
        self.la_cov_objective(x)
        self.theta = x[:self.R].copy()
        if _has_var_par(self.fl):
            self.sigma = x[self.R]

    def la_optim_bcov(self):
        x0 = np.r_[self.beta, self.theta]; lo = np.r_[np.full(self.P, -np.inf), np.full(self.R, 1e-6)]
        if _is_gaussian(self.fl):
            x0 = np.r_[x0, self.sigma]; lo = np.r_[lo, 0.0]
        x, _ = _minimise(self.la_btheta_objective, x0, lo)
        self.la_btheta_objective(x)
        self.beta = x[:self.P].copy(); self.theta = x[self.P:self.P + self.R].copy()
        if _is_gaussian(self.fl):
            self.sigma = x[self.P + self.R]

    def hess_la(self, tol=1e-4):
        x = np.r_[self.beta, self.theta]
        if _has_var_par(self.fl):
            x = np.r_[x, self.sigma]
        return optimhess(self.la_btheta_objective, x, tol)

    # mcmloptim.h:238-293
    def mcnr_b(self):
        zd = self.ZL @ self.v
        eta = self.xb + zd
        dmu = _vec(orc.lib().orc_detadmu, eta, self.lc)
        M = self.ZL.T @ (self.W[:, None] * self.ZL) + np.eye(self.Q)
        resid = self.y - _vec(orc.lib().orc_mod_inv, eta, self.lc)
        sigmas = np.sqrt(np.sum((resid - resid.mean()) ** 2) / (resid.size - 1))
        Wu = self.W * dmu * resid
        XtWX = self.X.T @ (self.W[:, None] * self.X)
        bincr = np.linalg.solve(XtWX, self.X.T @ Wu)
        # log_grad(v, usezl = false): mu = xb + Z v, grad = -D v + ZL'(score(mu)) * post
        vgrad = orc.log_grad(self.xb + self.Z @ self.v, self.ZL, self.y, self.var_par, self.fl,
                             np.zeros(self.Q)) - self.D0 @ self.v
        vincr = np.linalg.solve(M, vgrad)
        self.v = self.v + vincr
        self.beta = self.beta + bincr
        self.sigma = sigmas


def _driver(cov, data, eff_range, Z, X, y, family, link, start, usehess, tol, maxiter, nr):
    m = LaModel(cov, data, eff_range, Z, X, y, family, link, start)
    fl = m.fl
    beta, theta, var_par = m.beta.copy(), m.theta.copy(), 1.0
    if nr:
        m.update_W(True)                                            # src/mcml_la.cpp:195
    it, maxdiff, converged = 1, 1.0, False
    while maxdiff > tol and it <= maxiter:
        if nr:
            m.mcnr_b()
        else:
            m.la_optim()
        newbeta = m.beta.copy()
        m.xb = m.X @ newbeta
        m.update_W(bool(nr))
        m.la_optim_cov()
        newtheta = m.theta.copy()
        new_var_par = var_par
        if _is_gaussian(fl) or (nr and fl == 12):                   # :84 vs :222
            new_var_par = m.sigma
        maxdiff = max(np.max(np.abs(beta - newbeta)), np.max(np.abs(theta - newtheta)), abs(var_par - new_var_par))
        converged = maxdiff < tol
        beta, theta, var_par = newbeta, newtheta, new_var_par
        if not converged:
            m.L = orc.gen_D(cov, data, eff_range, theta, chol=True)
            m.xb = m.X @ beta
            if nr:
                m.var_par = new_var_par
                m.update_W(True)
            else:
                m.update_W(False)
                m.var_par = new_var_par
            m.ZL = m.Z @ m.L
        it += 1
    m.la_optim_bcov()
    beta, theta = m.beta.copy(), m.theta.copy()
    if _is_gaussian(fl):
        var_par = m.sigma
    se = np.zeros(np.asarray(start).size)
    if usehess:
        H = m.hess_la()
        Hi = np.linalg.inv(H)
        se[:H.shape[0]] = np.sqrt(np.diag(Hi))
    return dict(beta=beta, theta=theta, sigma=var_par, se=se, u=m.L @ m.v, v=m.v.copy(), converged=converged,
                iters=it - 1)


def mcml_la(cov, data, eff_range, Z, X, y, family, link, start, usehess=False, tol=1e-3, maxiter=10):
    """src/mcml_la.cpp:28-155"""
    return _driver(cov, data, eff_range, Z, X, y, family, link, start, usehess, tol, maxiter, False)


def mcml_la_nr(cov, data, eff_range, Z, X, y, family, link, start, usehess=False, tol=1e-3, maxiter=10):
    """src/mcml_la.cpp:174-290"""
    return _driver(cov, data, eff_range, Z, X, y, family, link, start, usehess, tol, maxiter, True)
