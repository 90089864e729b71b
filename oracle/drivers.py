"""Oracle-side drivers (TEST INFRASTRUCTURE ONLY): the reference's step exports
restated over the C oracle's objective pieces, with scipy doing the optimisation.

PARITY UNPINNED for the optimiser: rminqa's BOBYQA (mcmloptim.h:56-113) is not in
the image, so these drivers find the optimum of the SAME objective functors
(likelihood.h:31-110) with an independent, tightly converged optimiser; the
product's own BOBYQA must land on the same optimum.
"""
import numpy as np
from scipy import optimize

from . import oracle as orc


def _is_gaussian(fl):
    return fl in (7, 8)


def _minimise(fun, x0, lower):
    """independent bound-constrained minimiser, converged far below the 1e-6 the tests ask"""
    x0 = np.asarray(x0, float)
    bounds = [(lo if np.isfinite(lo) else None, None) for lo in lower]
    best = None
    for method, opts in (("COBYQA", dict(final_tr_radius=1e-10, maxfev=20000)),
                         ("Nelder-Mead", dict(xatol=1e-11, fatol=1e-14, maxiter=40000, maxfev=40000,
                                              adaptive=True))):
        try:
            r = optimize.minimize(fun, x0 if best is None else best.x, method=method, bounds=bounds, options=opts)
        except (ValueError, KeyError):
            continue
        if best is None or r.fun <= best.fun:
            best = r
    return best.x, best.fun


class Model:
    """the arguments every export receives"""

    def __init__(self, cov, data, eff_range, Z, X, y, family, link):
        self.cov, self.data, self.eff = cov, data, eff_range
        self.Z = np.asarray(Z, float); self.X = np.asarray(X, float); self.y = np.asarray(y, float)
        self.family, self.link = family, link
        self.fl = orc.flink(family, link)
        self.P = self.X.shape[1]
        self.R = orc.cov_npar(cov)

    # likelihood.h:57-64 / 40-45 / 88-109
    def L_obj(self, u, niter):
        def f(par):
            vp = par[self.P] if _is_gaussian(self.fl) else 0.0
            return -orc.model_loglik(self.Z, self.X @ par[:self.P], self.y, u, vp, self.fl, ncols=niter)
        return f

    def _mvn(self, th, u):
        try:
            return orc.mvn_ll(self.cov, self.data, self.eff, th, u)
        except RuntimeError:            # not positive definite: no likelihood there
            return -np.inf

    def D_obj(self, u):
        return lambda th: -self._mvn(th, u)

    def F_obj(self, u, niter, fix_var_par):
        def f(par):
            ll = orc.model_loglik(self.Z, self.X @ par[:self.P], self.y, u, fix_var_par, self.fl, ncols=niter)
            return -(ll + self._mvn(par[self.P:self.P + self.R], u))
        return f


def mcml_optim(mod, u, start, mcnr=False, niter=None):
    """src/mcml_optim.cpp:35-68"""
    start = np.asarray(start, float)
    niter = u.shape[1] if niter is None else niter
    beta = start[:mod.P].copy(); theta = start[mod.P:mod.P + mod.R].copy()
    sigma = start[mod.P + mod.R] if _is_gaussian(mod.fl) else 0.0
    if not mcnr:
        x0 = np.r_[beta, sigma] if _is_gaussian(mod.fl) else beta
        lo = np.r_[np.full(mod.P, -np.inf), 0.0] if _is_gaussian(mod.fl) else np.full(mod.P, -np.inf)
        x, _ = _minimise(mod.L_obj(u, niter), x0, lo)
        beta = x[:mod.P]
        if _is_gaussian(mod.fl):
            sigma = x[mod.P]
    else:
        r = orc.mcnr(mod.X, mod.Z, mod.y, u, beta, 1.0, mod.family, mod.link, ncols=niter)
        beta, sigma = r["beta"], r["sigma"]
    theta, _ = _minimise(mod.D_obj(u), theta, np.full(mod.R, 1e-6))
    return dict(beta=beta, theta=theta, sigma=sigma)


def mcml_simlik(mod, u, start):
    """src/mcml_optim.cpp:90-117 with the importance ratio evaluated in logs (defect D4)"""
    start = np.asarray(start, float)
    sigma = start[mod.P + mod.R] if _is_gaussian(mod.fl) else 0.0
    f = mod.F_obj(u, u.shape[1], sigma)
    x0 = start[:mod.P + mod.R]
    lo = np.r_[np.full(mod.P, -np.inf), np.full(mod.R, 1e-6)]
    x, _ = _minimise(f, x0, lo)
    return dict(beta=x[:mod.P], theta=x[mod.P:], sigma=sigma)


def optimhess(f, x, ndeps, lower=None, upper=None):
    """R's optimhess with a numerical gradient (optim.c), as rminqa's Functor::Hessian"""
    x = np.asarray(x, float); n = x.size
    lower = np.full(n, -np.inf) if lower is None else lower
    upper = np.full(n, np.inf) if upper is None else upper

    def grad(p):
        df = np.zeros(n)
        for i in range(n):
            eps = epsused = ndeps
            q = p.copy()
            tmp = p[i] + eps
            if tmp > upper[i]:
                tmp = upper[i]; epsused = tmp - p[i]
            q[i] = tmp; v1 = f(q)
            tmp = p[i] - eps
            if tmp < lower[i]:
                tmp = lower[i]; eps = p[i] - tmp
            q[i] = tmp; v2 = f(q)
            df[i] = (v1 - v2) / (epsused + eps)
        return df
    H = np.zeros((n, n))
    d = x.copy()
    for i in range(n):
        d[i] = d[i] + ndeps; g1 = grad(d)
        d[i] = d[i] - 2 * ndeps; g2 = grad(d)
        H[i, :] = (g1 - g2) / (2 * ndeps)
        d[i] = d[i] + ndeps
    return 0.5 * (H + H.T)


def mcml_hess(mod, u, start, tol=1e-5):
    """src/mcml_optim.cpp:263-285"""
    start = np.asarray(start, float)
    sigma = start[mod.P + mod.R] if _is_gaussian(mod.fl) else 0.0
    f = mod.F_obj(u, u.shape[1], sigma)
    lo = np.r_[np.full(mod.P, -np.inf), np.full(mod.R, 1e-6)]
    return optimhess(f, start[:mod.P + mod.R], tol, lo, np.full(mod.P + mod.R, np.inf))


def aic_mcml(mod, u, beta_par, cov_par):
    """src/mcml_optim.cpp:356-392"""
    beta_par = np.asarray(beta_par, float)
    var = mod.fl in (7, 8, 12)
    vp = beta_par[mod.P] if var else 0.0
    ll = orc.model_loglik(mod.Z, mod.X @ beta_par[:mod.P], mod.y, u, vp, mod.fl)
    return -2 * (ll + orc.mvn_ll(mod.cov, mod.data, mod.eff, cov_par, u)) + 2 * (beta_par.size + len(cov_par))


def sample(mod, beta, theta, var_par, warmup, m, lambda_, maxsteps, target_accept, seed, iter_idx, chains,
           chain_offset=0):
    """the sampler as the build defines it: `chains` chains (global ids chain_offset..), reference
    layout when chains == 1 (mhmcmc.h:121-157)"""
    L = orc.gen_D(mod.cov, mod.data, mod.eff, theta, chol=True)
    ZL = mod.Z @ L
    xb = mod.X @ beta
    if chains <= 1:
        s, _, _, _ = orc.hmc_chain(xb, ZL, mod.y, var_par, mod.fl, warmup, m, lambda_, maxsteps, target_accept,
                                   seed, chain_id=chain_offset, iter_idx=iter_idx)
        return L @ s, m
    d = -(-m // chains)
    cols = []
    for c in range(chains):
        s, _, _, _ = orc.hmc_chain(xb, ZL, mod.y, var_par, mod.fl, warmup, d, lambda_, maxsteps, target_accept,
                                   seed, chain_id=chain_offset + c, iter_idx=iter_idx)
        cols.append(s[:, 1:])
    u = L @ np.concatenate(cols, axis=1)
    return u, u.shape[1]
