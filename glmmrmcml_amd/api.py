"""Host-side mirror of glmmrMCML's Rcpp export surface (R/RcppExports.R:35-303)
over the C ABI of libglmmr_mcml_hip.so.  R is not in the image, so this module
stands where ``R6ModelExtMCML.R`` does: same export names, same argument order
and meaning, numpy arrays instead of R vectors.  Everything is converted to
column-major float64 / int32 and handed over as plain pointers; no torch types
cross the boundary.
"""
import ctypes as C

import numpy as np

from . import _lib

c_dp = C.POINTER(C.c_double)
c_ip = C.POINTER(C.c_int32)

REDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int)


class Problem(C.Structure):
    _fields_ = [("cov", c_ip), ("cov_rows", C.c_int), ("data", c_dp), ("data_len", C.c_int),
                ("eff_range", c_dp), ("eff_len", C.c_int), ("Z", c_dp), ("X", c_dp), ("y", c_dp),
                ("n", C.c_int), ("Q", C.c_int), ("P", C.c_int), ("family", C.c_char_p),
                ("link", C.c_char_p)]


class DevOpts(C.Structure):
    _fields_ = [("device", C.c_int), ("stream", C.c_void_p), ("rank", C.c_int), ("world", C.c_int),
                ("reduce", REDUCE_FN), ("reduce_user", C.c_void_p)]


class HmcOpts(C.Structure):
    _fields_ = [("warmup", C.c_int), ("nsamp", C.c_int), ("adapt", C.c_int), ("lambda_", C.c_double),
                ("max_steps", C.c_int), ("target_accept", C.c_double), ("chains", C.c_int),
                ("chain_offset", C.c_int)]


class HmcDiag(C.Structure):
    _fields_ = [("accept_rate", C.c_double), ("mean_e", C.c_double), ("min_e", C.c_double),
                ("max_e", C.c_double), ("max_steps_used", C.c_int), ("leapfrog_total", C.c_longlong)]


def _f(a):
    return np.asfortranarray(np.asarray(a, dtype=np.float64))


def _i(a):
    return np.asfortranarray(np.asarray(a, dtype=np.int32))


def _p(a):
    return None if a is None else a.ctypes.data_as(c_dp)


class Context:
    """A device-resident MCML problem (glmmr_mcml_ctx).  Inputs are uploaded once;
    afterwards only parameter vectors go down and scalars come back."""

    def __init__(self, cov, data, eff_range, Z=None, X=None, y=None, family=None, link=None,
                 device=0, stream=None, rank=0, world=1, reduce=None):
        L = _lib.lib()
        self._cov = _i(cov)
        assert self._cov.ndim == 2 and self._cov.shape[1] == 5, "cov must be rows x 5"
        self._data = _f(data).ravel()
        self._eff = _f(eff_range).ravel()
        p = Problem()
        p.cov = self._cov.ctypes.data_as(c_ip); p.cov_rows = self._cov.shape[0]
        p.data = _p(self._data); p.data_len = self._data.size
        p.eff_range = _p(self._eff); p.eff_len = self._eff.size
        self.n = self.P = 0
        if Z is not None:
            self._Z = _f(Z); self._X = _f(X); self._y = _f(y).ravel()
            self.n, self.Q = self._Z.shape
            self.P = self._X.shape[1]
            assert self._X.shape[0] == self.n and self._y.size == self.n
            p.Z = _p(self._Z); p.X = _p(self._X); p.y = _p(self._y)
            p.n = self.n; p.Q = self.Q; p.P = self.P
            p.family = family.encode(); p.link = link.encode()
        o = DevOpts()
        o.device = device; o.stream = stream; o.rank = rank; o.world = world
        self._reduce_cb = REDUCE_FN(reduce) if reduce is not None else REDUCE_FN()
        o.reduce = self._reduce_cb
        self._h = C.c_void_p()
        _lib.check(L.glmmr_mcml_ctx_create(C.byref(p), C.byref(o), C.byref(self._h)))
        self.family, self.link = family, link
        if Z is None:
            self.Q = int(sum(self._cov[np.r_[True, self._cov[1:, 0] != self._cov[:-1, 0]], 1]))
        self.mcols = 0

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            _lib.lib().glmmr_mcml_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- samples
    def set_u(self, u, niter=None):
        u = _f(u)
        if u.ndim == 1:
            u = u.reshape(-1, 1, order="F")
        self.mcols = u.shape[1]
        _lib.check(_lib.lib().glmmr_mcml_set_u(self._h, _p(u), u.shape[0], u.shape[1],
                                               u.shape[1] if niter is None else niter))

    def get_u(self):
        u = np.zeros((self.Q, self.mcols), order="F")
        _lib.check(_lib.lib().glmmr_mcml_get_u(self._h, _p(u), self.Q))
        return u

    # -- A8
    def mvn_ll(self, theta):
        theta = _f(theta).ravel()
        out = C.c_double()
        _lib.check(_lib.lib().glmmr_mcml_ctx_mvn_ll(self._h, _p(theta), C.byref(out)))
        return out.value

    def gen_D(self, theta, chol=False):
        theta = _f(theta).ravel()
        D = np.zeros((self.Q, self.Q), order="F")
        _lib.check(_lib.lib().glmmr_mcml_ctx_gen_D(self._h, _p(theta), int(chol), _p(D), self.Q))
        return D

    # -- A6 / A7
    def loglik(self, beta, var_par):
        beta = _f(beta).ravel()
        out = C.c_double()
        _lib.check(_lib.lib().glmmr_mcml_ctx_loglik(self._h, _p(beta), C.c_double(var_par), C.byref(out)))
        return out.value

    def mcnr(self, beta, var_par):
        beta = _f(beta).ravel()
        bout = np.zeros(self.P); sig = C.c_double()
        stats = np.zeros(self.P * self.P + self.P + 1)
        _lib.check(_lib.lib().glmmr_mcml_ctx_mcnr(self._h, _p(beta), C.c_double(var_par), _p(bout),
                                                  C.byref(sig), _p(stats)))
        P = self.P
        return dict(beta=bout, sigma=sig.value, XtWX=stats[:P * P].reshape(P, P, order="F"),
                    XtWr=stats[P * P:P * P + P], sigma_sum=stats[-1])

    # -- model state / sampler
    def update_L(self, theta):
        theta = _f(theta).ravel()
        _lib.check(_lib.lib().glmmr_mcml_ctx_update_L(self._h, _p(theta)))

    def set_L(self, Lmat):
        Lmat = _f(Lmat)
        _lib.check(_lib.lib().glmmr_mcml_ctx_set_L(self._h, _p(Lmat), Lmat.shape[0]))

    def log_prob_grad(self, beta, var_par, V):
        """log_prob / log_grad of every column of V (test hook for A4/A5)"""
        beta = _f(beta).ravel(); V = _f(V)
        lp = np.zeros(V.shape[1]); G = np.zeros_like(V, order="F")
        _lib.check(_lib.lib().glmmr_mcml_dbg_log_prob_grad(self._h, _p(beta), C.c_double(var_par), _p(V),
                                                           V.shape[1], _p(lp), _p(G)))
        return lp, G

    def hmc_sample(self, beta, var_par, warmup, nsamp, lambda_, max_steps, target_accept, seed,
                   chains=1, chain_offset=0, iter_idx=0, adapt=100, inj_init=None, inj_mom=None,
                   want_trace=False):
        beta = _f(beta).ravel()
        o = HmcOpts(warmup, nsamp, adapt, lambda_, max_steps, target_accept, chains, chain_offset)
        d = HmcDiag()
        ii = None if inj_init is None else _f(inj_init)
        im = None if inj_mom is None else _f(inj_mom)
        total = warmup + (nsamp if chains == 1 else -(-nsamp // chains))   # proposals per chain
        flags = np.zeros((chains, total), dtype=np.uint8, order="F") if want_trace else None
        probs = np.zeros((chains, total), order="F") if want_trace else None
        ncols = C.c_int()
        _lib.check(_lib.lib().glmmr_mcml_ctx_hmc_sample(
            self._h, _p(beta), C.c_double(var_par), C.byref(o), C.c_uint64(seed), C.c_uint32(iter_idx),
            _p(ii), _p(im), None if flags is None else flags.ctypes.data_as(C.POINTER(C.c_uint8)),
            _p(probs), C.byref(d), C.byref(ncols)))
        self.mcols = ncols.value
        diag = dict(accept_rate=d.accept_rate, mean_e=d.mean_e, min_e=d.min_e, max_e=d.max_e,
                    max_steps_used=d.max_steps_used, leapfrog_total=d.leapfrog_total)
        if want_trace:
            return diag, flags, probs
        return diag


# ---------------------------------------------------------------------------
# Mirrors of the Rcpp exports
# ---------------------------------------------------------------------------
def mvn_ll(cov, data, eff_range, gamma, u):
    """mvn_ll(cov, data, eff_range, gamma, u)  -- src/mcml_optim.cpp:406-414"""
    cov = _i(cov); data = _f(data).ravel(); eff = _f(eff_range).ravel(); gamma = _f(gamma).ravel()
    u = _f(u)
    if u.ndim == 1:
        u = u.reshape(-1, 1, order="F")
    out = C.c_double()
    _lib.check(_lib.lib().glmmr_mcml_mvn_ll(cov.ctypes.data_as(c_ip), cov.shape[0], _p(data), data.size,
                                            _p(eff), eff.size, _p(gamma), gamma.size, _p(u), u.shape[0],
                                            u.shape[1], C.byref(out)))
    return out.value
