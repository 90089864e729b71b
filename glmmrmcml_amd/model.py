"""ModelMCML -- the caller side of the hot path: what `ModelMCML$MCML()` and `ModelMCML$LA()` do around the
exported C++ functions (R/R6ModelExtMCML.R:103-594 and 627-845), restated on top of glmmrmcml_amd.api.

glmmrBase's Covariance / MeanFunction objects are not in the image, so the model is described by what they hand
to the exports: the `get_D_data()` triple (cov, data, eff_range), Z, X, the family / link strings and the stored
parameter values.  Everything here is host-side bookkeeping: start-vector assembly and checks, the parInds index
sets, the choice of export per option, standard errors, cAIC, the approximate R-squared and the `mcml` result
record (plus its print method, R/printfunctions.R:20-70).

Differences from the R class, each deliberate:
  * usestan = TRUE needs cmdstanr + Stan, which do not exist here: the same outer loop (R6ModelExtMCML.R:236-338:
    sample u, call mcml_optim[_sparse], rebuild L) runs with the package's own sampler export `mcmc_sample` in
    Stan's place when `sampler="stepwise"`; the default `sampler="full"` is the usestan = FALSE branch (one call of
    `mcml_full`).
  * se.method "lik" / "robust": the R code assigns the dense Hessian to `newtheta` and then inverts an undefined
    `hess` (defect D6), and never forwards `fd_tol` (D12); here the Hessian returned by mcml_hess is the one
    inverted and `fd_tol` is its step.
  * `mcml_simlik` is called without the non-existent `mcnr=` argument (defect D7).
  * `information_matrix()` lives in glmmrBase; its GLS form X' (W^-1 + Z D Z')^-1 X is restated from the
    commented block of src/mcml_la.cpp:126-139 (INFERRED, unverifiable without glmmrBase).
"""
import math

import numpy as np

from . import api as _api

_FNPAR = (1, 1, 1, 2, 2, 1, 2, 2, 2, 2, 2, 2, 2, 1)          # parameters per covariance function id 1..14 (:430)
_Z975 = 1.959963984540054                                     # qnorm(1 - 0.05 / 2)
_VAR_FAMILIES = ("gaussian", "Gamma", "beta")


_FLINK = {"poissonlog": 1, "poissonidentity": 2, "binomiallogit": 3, "binomiallog": 4, "binomialidentity": 5,
          "binomialprobit": 6, "gaussianidentity": 7, "gaussianlog": 8, "gammalog": 9, "gammainverse": 10,
          "gammaidentity": 11, "betalogit": 12}                # mcmlmodel.h:74-87


def _dhdmu(xb, family, link):
    """glmmrBase `gen_dhdmu` (R6ModelExtMCML.R:558,815) is the Rcpp export of `glmmr::maths::dhdmu`, the function
    mcmlmodel.h:122 calls for the MCNR weights: ONE table, the one csrc/glm.h::glm_dhdmu and
    oracle/mcml_oracle.c::orc_dhdmu restate.  glmmrBase is not in the image, so the table is INFERRED (parity
    unpinned): it is written the way glmmrBase 0.2.x's maths.h is remembered to write it -- cases 3, 4, 5 and 12
    take p from the LOGISTIC inverse link whatever the link, case 8 is exp(-eta) -- and not re-derived from GLM theory
    (for binomial/log and binomial/identity the textbook working weights differ).  tests/test_model_caller.py
    checks this function against the oracle's for all 12 cases."""
    xb = np.asarray(xb, float)
    key = family.lower() + link
    if key not in _FLINK:
        raise ValueError("unknown family/link %s/%s" % (family, link))
    fl = _FLINK[key]
    if fl == 1:
        return np.exp(-1.0 * xb)
    if fl == 2:
        return np.exp(xb)
    if fl in (3, 4, 5, 12):
        p = np.exp(xb) / (1 + np.exp(xb))
        if fl == 4:
            return (1.0 - p) / p
        if fl == 5:
            return p * (1.0 - p)
        return 1 / (p * (1.0 - p))
    if fl == 6:
        from math import erfc
        p = np.array([0.5 * erfc(-v * 0.70710678118654752440) for v in np.atleast_1d(xb)]).reshape(xb.shape)
        d = np.exp(-0.5 * xb * xb) * 0.39894228040143267794
        return (p * (1 - p)) / d
    if fl in (7, 9):
        return np.ones_like(xb)
    if fl == 8:
        return 1 / np.exp(xb)
    if fl == 10:
        return 1 / (xb * xb)
    return xb * xb                                             # 11


class McmlFit(dict):
    """the `mcml` list MCML() / LA() return (R6ModelExtMCML.R:571-587); attribute access for convenience"""

    __getattr__ = dict.__getitem__

    def table(self):
        """rows of print.mcml (R/printfunctions.R:41-58): (name, est, se, z, p, lower, upper) without the d's"""
        co = self["coefficients"]
        nd = self["re_samps"].shape[0]
        rows = []
        k = len(co["par"]) - nd
        names = list(co["par"][:k])
        for nm in set(names):                                  # duplicated names get .1, .2, ...
            idx = [i for i, x in enumerate(names) if x == nm]
            if len(idx) > 1:
                for j, i in enumerate(idx):
                    names[i] = "%s.%d" % (nm, j + 1)
        for i in range(k):
            est, se = co["est"][i], co["SE"][i]
            z = est / se if se and not math.isnan(se) else float("nan")
            p = 2 * (1 - 0.5 * (1 + math.erf(abs(z) / math.sqrt(2)))) if not math.isnan(z) else float("nan")
            rows.append((names[i], est, se, z, p, co["lower"][i], co["upper"][i]))
        return rows

    def __str__(self):
        m = self["method"]
        head = ("Markov chain Monte Carlo Maximum Likelihood Estimation\nAlgorithm: " if m in ("mcem", "mcnr")
                else "Maximum Likelihood Estimation with Laplace Approximation\nAlgorithm: ")
        alg = {"nr": "Newton-Raphson", "nloptim": "BOBYQA", "mcem": "Markov Chain Expectation Maximisation",
               "mcnr": "Markov Chain Newton-Raphson"}[m]
        out = [head + alg + (" with simulated likelihood step" if self["sim_lik"] else ""),
               "Family: %s , Link function: %s" % (self["family"], self["link"])]
        if m in ("mcem", "mcnr"):
            out.append("Number of Monte Carlo simulations per iteration: %s with tolerance %s" % (self["m"], self["tol"]))
        out.append("%-14s %10s %10s %8s %8s %10s %10s" % ("", "Estimate", "Std. Err.", "z value", "p value", "2.5% CI",
                                                         "97.5% CI"))
        for r in self.table():
            out.append("%-14s %10.2f %10.2f %8.2f %8.2f %10.2f %10.2f" % r)
        out.append("cAIC: %.2f" % self["aic"])
        out.append("Approximate R-squared: Conditional: %.2f  Marginal: %.2f" % (self["Rsq"]["cond"], self["Rsq"]["marg"]))
        if not self["converged"]:
            out.append("Warning: algorithm did not converge")
        return "\n".join(out)


class ModelMCML:
    """covariance = the get_D_data() triple + Z + stored parameters; mean = X + family/link + stored parameters"""

    def __init__(self, cov, data, eff_range, Z, X, family, link, mean_parameters, cov_parameters, var_par=1.0,
                 x_names=None, cov_names=None, backend=None):
        self.cov = np.asfortranarray(np.asarray(cov, dtype=np.int32))
        self.data = np.asarray(data, float).ravel()
        self.eff_range = np.asarray(eff_range, float).ravel()
        self.Z = np.asfortranarray(np.asarray(Z, float))
        self.X = np.asfortranarray(np.asarray(X, float))
        self.family, self.link = family, link
        self.mean_parameters = np.asarray(mean_parameters, float).ravel().copy()
        self.cov_parameters = np.asarray(cov_parameters, float).ravel().copy()
        self.var_par = float(var_par)
        self.x_names = list(x_names) if x_names is not None else ["b%d" % (i + 1) for i in range(self.X.shape[1])]
        self.cov_names = list(cov_names) if cov_names is not None else None
        # R6ModelExtMCML.R:868-873
        self.mcmc_options = dict(warmup=500, samps=250, lambda_=5.0, refresh=500, maxsteps=100, target_accept=0.95)
        self._be = backend if backend is not None else _api      # tests inject a recording backend
        if self.X.shape[0] != self.Z.shape[0]:
            raise ValueError("X and Z have different numbers of rows")
        if self.mean_parameters.size != self.X.shape[1]:
            raise ValueError("wrong number of mean function parameters")

    # ---- helpers -------------------------------------------------------------------------------------------------
    def n(self):
        return self.X.shape[0]

    def _check_y(self, y):
        """R6ModelExtMCML.R:133-143 / 637-647"""
        y = np.asarray(y, float).ravel()
        if y.size != self.n():
            raise ValueError("y has the wrong length")
        f = self.family
        if f == "binomial" and not np.all((y == 0) | (y == 1)):
            raise ValueError("y must be 0 or 1")
        if f == "poisson" and (np.any(y < 0) or np.any(y % 1 != 0)):
            raise ValueError("y must be integer >= 0")
        if f == "beta" and (np.any(y < 0) or np.any(y > 1)):
            raise ValueError("y must be between 0 and 1")
        if f == "Gamma" and np.any(y <= 0):
            raise ValueError("y must be positive")
        if f == "gaussian" and self.link == "log" and np.any(y <= 0):
            raise ValueError("y must be positive")
        return y

    def _start(self, start):
        """start vector assembly (R6ModelExtMCML.R:160-181): always P + R + 1 long when it reaches C++"""
        P, R = self.X.shape[1], self.cov_parameters.size
        if self.family in _VAR_FAMILIES:
            if start is None:
                start = np.r_[self.mean_parameters, self.cov_parameters, self.var_par]
            start = np.asarray(start, float).ravel()
            if start.size != P + R + 1:
                raise ValueError("wrong number of starting values")
            all_pars = np.arange(P + R + 1)
        elif self.family in ("binomial", "poisson"):
            if start is None:
                start = np.r_[self.mean_parameters, self.cov_parameters]
            start = np.asarray(start, float).ravel()
            if start.size != P + R:
                raise ValueError("wrong number of starting values")
            start = np.r_[start, 1.0]
            all_pars = np.arange(P + R)
        else:
            raise ValueError("family %r is not supported" % (self.family,))
        par_inds = dict(b=np.arange(P), cov=np.arange(P, P + R), sig=P + R)
        return start, all_pars, par_inds

    def _ddata(self):
        return (self.cov, self.data, self.eff_range)

    def _cov_par_names(self):
        """cov_pars_names (R6ModelExtMCML.R:427-440): the random-effect term names repeated by the number of
        parameters of their functions (fnpar); callers without term labels get cov1..covR"""
        R = self.cov_parameters.size
        if self.cov_names is not None:
            if len(self.cov_names) == R:
                return list(self.cov_names)
            seen = []
            for r in range(self.cov.shape[0]):                  # (function id, first parameter index) per term
                key = (int(self.cov[r, 2]), int(self.cov[r, 4]))
                if key not in seen:
                    seen.append(key)
            names = []
            for term, (fid, _) in enumerate(sorted(seen, key=lambda k: k[1])):
                names += [self.cov_names[min(term, len(self.cov_names) - 1)]] * _FNPAR[fid - 1]
            if len(names) == R:
                return names
        return ["cov%d" % (i + 1) for i in range(R)]

    def information_matrix(self, theta_cov, beta, sigma):
        """GLS information X' Sigma^-1 X, Sigma = W^-1 + Z D Z' (see module docstring; INFERRED)"""
        with self._be.Context(self.cov, self.data, self.eff_range) as ctx:
            D = ctx.gen_D(np.asarray(theta_cov, float))
        xb = self.X @ beta
        w = _dhdmu(xb, self.family, self.link)
        if self.family == "gaussian":
            w = w * sigma * sigma
        elif self.family == "Gamma":
            w = w * sigma
        elif self.family == "beta":
            w = w * (1 + sigma)
        S = np.diag(w) + self.Z @ D @ self.Z.T
        return self.X.T @ np.linalg.solve(S, self.X)

    def _rsq(self, theta, par_inds, zd):
        """approximate R-squared (R6ModelExtMCML.R:555-569)"""
        xb = self.X @ theta[par_inds["b"]]
        wdiag = _dhdmu(xb, self.family, self.link)
        if self.family in ("gaussian", "gamma"):                # sic: lower-case "gamma" never matches "Gamma"
            wdiag = theta[par_inds["sig"]] * wdiag
        vx, vz = np.var(xb, ddof=1), np.var(zd, ddof=1)
        total = vx + vz + np.mean(wdiag)
        return dict(cond=(vx + vz) / total, marg=vx / total)

    def _coef_table(self, names, est, SE):
        est = np.asarray(est, float); SE = np.asarray(SE, float)
        return dict(par=list(names), est=est, SE=SE, lower=est - _Z975 * SE, upper=est + _Z975 * SE)

    # ---- MCML ----------------------------------------------------------------------------------------------------
    def MCML(self, y, start=None, se_method="approx", method="mcnr", sim_lik_step=False, verbose=True, tol=1e-2,
             max_iter=30, sparse=False, sampler="full", options=None, seed=0, chains=1):
        """ModelMCML$MCML (R6ModelExtMCML.R:103-594).  sampler: "full" = the usestan = FALSE branch (mcml_full);
        "stepwise" = the usestan = TRUE loop with mcmc_sample standing in for Stan, "nuts" = the same loop with
        gen_u_samples (the build's No-U-Turn sampler, csrc/nuts.h) in Stan's place.  seed / chains are the build's
        additions (0 = random_device, 1 chain = the reference)."""
        if se_method not in ("lik", "robust", "approx", "none"):
            raise ValueError("se.method should be 'lik', 'robust', 'approx', or 'none'")
        if method not in ("mcem", "mcnr"):
            raise ValueError("method should be 'mcem' or 'mcnr'")
        options = {} if options is None else options
        if not isinstance(options, dict):
            raise ValueError("options should be a list")
        no_warnings = bool(options.get("no_warnings", False))
        fd_tol = float(options.get("fd_tol", 1e-4))
        trace = int(options.get("trace", 0))
        maxfun = int(options.get("maxfun", 0))
        y = self._check_y(y)
        P, R = self.X.shape[1], self.cov_parameters.size
        start, all_pars, par = self._start(start)
        mf_par = np.r_[par["b"], par["sig"]] if self.family in _VAR_FAMILIES else par["b"]
        theta = start.copy()
        be, mo = self._be, self.mcmc_options
        Q = self.Z.shape[1]
        warnings = []
        pattern = None
        if sparse:
            pattern = self._sparse_pattern()
        it = 0
        if sampler == "full":
            if sparse:
                raise ValueError("sparse = TRUE is only reached through the stepwise loop (mcml_optim_sparse)")
            res = be.mcml_full(*self._ddata(), self.Z, self.X, y, self.family, self.link, theta.copy(), mcnr=(method == "mcnr"),
                               m=mo["samps"], maxiter=max_iter, warmup=mo["warmup"], tol=tol, verbose=verbose,
                               lambda_=mo["lambda_"], trace=trace, refresh=mo["refresh"], maxsteps=mo["maxsteps"],
                               target_accept=mo["target_accept"], seed=seed, chains=chains, maxfun=maxfun)
            theta[par["b"]] = res["beta"]
            if self.family in _VAR_FAMILIES:
                theta[par["sig"]] = res["sigma"]
            theta[par["cov"]] = res["theta"]
            not_conv = not res["converged"]
            dsamps = np.asarray(res["u"])
        elif sampler in ("stepwise", "nuts"):
            with be.Context(self.cov, self.data, self.eff_range) as ctx:
                L = ctx.gen_D(theta[par["cov"]], chol=True)
            thetanew = np.ones_like(theta)
            dsamps = None
            while np.any(np.abs(theta - thetanew) > tol) and it <= max_iter:
                it += 1
                thetanew = theta.copy()
                if sampler == "nuts":                            # mod$sample(...) + L %*% t(draws), :246-254
                    dsamps = be.gen_u_samples(y, self.X, self.Z, L, thetanew[par["b"]], self.family, self.link,
                                              sigma=thetanew[par["sig"]], warmup_iter=mo["warmup"], m=mo["samps"],
                                              seed=(seed + it if seed else 0), chains=chains)
                else:
                    dsamps = be.mcmc_sample(self.Z, L, self.X, y, thetanew[par["b"]], self.family, self.link, mo["warmup"],
                                            mo["samps"], mo["lambda_"], var_par=thetanew[par["sig"]], trace=trace,
                                            refresh=mo["refresh"], maxsteps=mo["maxsteps"],
                                            target_accept=mo["target_accept"], seed=(seed + it if seed else 0),
                                            chains=chains)
                if sparse:
                    fit = be.mcml_optim_sparse(*self._ddata(), pattern[0], pattern[1], self.Z, self.X, y, dsamps,
                                               self.family, self.link, theta.copy(), trace=trace, mcnr=(method == "mcnr"),
                                               maxfun=maxfun)
                else:
                    fit = be.mcml_optim(*self._ddata(), self.Z, self.X, y, dsamps, self.family, self.link, theta.copy(),
                                        trace=trace, mcnr=(method == "mcnr"), maxfun=maxfun)
                theta[par["b"]] = np.ravel(fit["beta"])
                if self.family in _VAR_FAMILIES:
                    theta[par["sig"]] = fit["sigma"]
                theta[par["cov"]] = np.ravel(fit["theta"])
                if sparse:
                    L = self._L_from_ldl(fit, Q)                 # SparseChol::sparse_L(fit) %*% diag(sqrt(D)), :313-315
                else:
                    with be.Context(self.cov, self.data, self.eff_range) as ctx:
                        L = ctx.gen_D(thetanew[par["cov"]], chol=True)   # sic: the PREVIOUS theta (:317)
                if verbose:
                    print("Iter %d  Beta: %s  Theta: %s  Max. diff: %g" % (it, theta[par["b"]], theta[par["cov"]],
                                                                          np.max(np.abs(theta - thetanew))))
            not_conv = it >= max_iter or bool(np.any(np.abs(theta - thetanew) > tol))
        else:
            raise ValueError("sampler should be 'full', 'stepwise' or 'nuts'")
        if not_conv and not no_warnings:
            warnings.append("algorithm not converged")
        if sim_lik_step:
            if sparse:
                new = be.mcml_simlik_sparse(*self._ddata(), pattern[0], pattern[1], self.Z, self.X, y, dsamps,
                                            self.family, self.link, theta.copy(), trace=trace, maxfun=maxfun)
            else:
                new = be.mcml_simlik(*self._ddata(), self.Z, self.X, y, dsamps, self.family, self.link, theta.copy(),
                                     trace=trace, maxfun=maxfun)
            newtheta = np.r_[np.ravel(new["beta"]), np.ravel(new["theta"])]
            if self.family in _VAR_FAMILIES:
                newtheta = np.r_[newtheta, new["sigma"]]
            theta[all_pars] = newtheta[:all_pars.size]

        # ---- standard errors (R6ModelExtMCML.R:427-527) ----
        cov_names = self._cov_par_names()
        hessused = False
        names = self.x_names + cov_names + (["sigma"] if self.family in _VAR_FAMILIES else [])
        if se_method in ("lik", "robust", "approx"):
            SE = np.full(P + R, np.nan)
            if se_method in ("lik", "robust"):
                try:
                    if sparse:
                        H = be.mcml_hess_sparse(*self._ddata(), pattern[0], pattern[1], self.Z, self.X, y, dsamps,
                                                self.family, self.link, theta.copy(), tol=fd_tol, trace=trace)
                    else:
                        H = be.mcml_hess(*self._ddata(), self.Z, self.X, y, dsamps, self.family, self.link, theta.copy(),
                                         tol=fd_tol, trace=trace)
                    hessused = True
                    with np.errstate(invalid="ignore"):
                        SE = np.sqrt(np.diag(np.linalg.inv(np.asarray(H))))[:P + R]
                except (np.linalg.LinAlgError, _api._lib.McmlError):
                    SE = np.full(P + R, np.nan)
            if se_method == "approx" or np.any(np.isnan(SE[:P])):
                SE = np.full(P + R, np.nan)
                hessused = False
                M = self.information_matrix(theta[par["cov"]], theta[par["b"]], theta[par["sig"]])
                SE[:P] = np.sqrt(np.diag(np.linalg.inv(M)))
            if self.family in _VAR_FAMILIES:
                SE = np.r_[SE, np.nan]
            coef = self._coef_table(names + ["d%d" % (i + 1) for i in range(Q)],
                                    np.r_[theta[all_pars], dsamps.mean(axis=1)],
                                    np.r_[SE, dsamps.std(axis=1, ddof=1) if dsamps.shape[1] > 1 else np.full(Q, np.nan)])
        else:
            coef = self._coef_table(names + ["d%d" % (i + 1) for i in range(Q)],
                                    np.r_[theta[all_pars], dsamps.mean(axis=1)], np.full(all_pars.size + Q, np.nan))
        aic = be.aic_mcml(*self._ddata(), self.Z, self.X, y, dsamps, self.family, self.link, theta[mf_par], theta[par["cov"]])
        rsq = self._rsq(theta, par, self.Z @ dsamps.mean(axis=1))
        return McmlFit(coefficients=coef, converged=not not_conv, method=method, hessian=hessused, m=mo["samps"], tol=tol,
                       sim_lik=bool(sim_lik_step), aic=float(aic), Rsq=rsq, family=self.family, link=self.link,
                       re_samps=dsamps, iter=it, warnings=warnings, theta=theta)

    # ---- LA ------------------------------------------------------------------------------------------------------
    def LA(self, y, start=None, method="nloptim", use_hess=False, verbose=False, maxfun=0):
        """ModelMCML$LA (R6ModelExtMCML.R:627-845): mcml_la ("nloptim") or mcml_la_nr ("nr"), tol fixed at 1e-2"""
        if method not in ("nloptim", "nr"):
            raise ValueError("method should be either nr or nloptim")
        trace = 1 if verbose else 0
        y = self._check_y(y)
        P, R = self.X.shape[1], self.cov_parameters.size
        start, all_pars, par = self._start(start)
        mf_par = np.r_[par["b"], par["sig"]] if self.family == "gaussian" else par["b"]     # :663-667
        theta = start.copy()
        fn = self._be.mcml_la if method == "nloptim" else self._be.mcml_la_nr
        resb = fn(*self._ddata(), self.Z, self.X, y, self.family, self.link, theta.copy(), usehess=use_hess, tol=1e-2,
                  verbose=verbose, trace=trace, maxfun=maxfun)
        theta[par["b"]] = resb["beta"]
        if self.family in _VAR_FAMILIES:
            theta[par["sig"]] = resb["sigma"]
        theta[par["cov"]] = resb["theta"]
        Q = self.Z.shape[1]
        u = np.asarray(resb["u"], float).reshape(Q, 1)
        names = self.x_names + self._cov_par_names() + (["sigma"] if self.family in _VAR_FAMILIES else [])
        if not use_hess:
            SE = np.full(P + R, np.nan)
            M = self.information_matrix(theta[par["cov"]], theta[par["b"]], theta[par["sig"]])
            SE[:P] = np.sqrt(np.diag(np.linalg.inv(M)))
            if self.family in _VAR_FAMILIES:
                SE = np.r_[SE, np.nan]
        else:
            SE = np.asarray(resb["se"], float)[:all_pars.size]
        coef = self._coef_table(names + ["d%d" % (i + 1) for i in range(Q)], np.r_[theta[all_pars], u.ravel()],
                                np.r_[SE, np.full(Q, np.nan)])
        aic = self._be.aic_mcml(*self._ddata(), self.Z, self.X, y, u, self.family, self.link, theta[mf_par],
                                theta[par["cov"]])
        xb = self.X @ theta[par["b"]]
        wdiag = _dhdmu(xb, self.family, self.link)
        if self.family in _VAR_FAMILIES:                       # :819-821 (here the three families, unlike MCML)
            wdiag = theta[par["sig"]] * wdiag
        zd = self.Z @ u.ravel()
        vx, vz = np.var(xb, ddof=1), np.var(zd, ddof=1)
        total = vx + vz + np.mean(wdiag)
        return McmlFit(coefficients=coef, converged=True, method=method, hessian=False, m=None, tol=None, sim_lik=False,
                       aic=float(aic), Rsq=dict(cond=(vx + vz) / total, marg=vx / total), family=self.family,
                       link=self.link, re_samps=u, iter=0, warnings=[], theta=theta)

    # ---- sparse helpers ------------------------------------------------------------------------------------------
    def _sparse_pattern(self):
        """Ap, Ai of D as a dsCMatrix (upper triangle, CSC) -- R6ModelExtMCML.R:193-195"""
        with self._be.Context(self.cov, self.data, self.eff_range) as ctx:
            D = ctx.gen_D(self.cov_parameters)
        Qn = D.shape[0]
        Ap, Ai = [0], []
        for j in range(Qn):
            rows = [i for i in range(j + 1) if D[i, j] != 0.0]
            Ai += rows
            Ap.append(len(Ai))
        return np.array(Ap, dtype=np.int32), np.array(Ai, dtype=np.int32)

    @staticmethod
    def _L_from_ldl(fit, Q):
        """SparseChol::sparse_L(fit) %*% Diagonal(sqrt(D)) (R6ModelExtMCML.R:313-315): unit-lower CSC without the
        diagonal (Ap, Ai, Ax) + pivots D"""
        Lm = np.eye(Q)
        Ap, Ai, Ax = fit["Ap"], fit["Ai"], fit["Ax"]
        for j in range(Q):
            for k in range(Ap[j], Ap[j + 1]):
                Lm[Ai[k], j] = Ax[k]
        return Lm * np.sqrt(np.asarray(fit["D"], float))[None, :]
