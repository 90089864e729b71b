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


class Ext(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("chains", C.c_int), ("maxfun", C.c_int), ("device", C.c_int),
                ("theta_batch", C.c_int)]


class HmcDiag(C.Structure):
    _fields_ = [("accept_rate", C.c_double), ("mean_e", C.c_double), ("min_e", C.c_double),
                ("max_e", C.c_double), ("max_steps_used", C.c_int), ("leapfrog_total", C.c_longlong)]


class NutsOpts(C.Structure):
    _fields_ = [("warmup", C.c_int), ("nsamp", C.c_int), ("max_treedepth", C.c_int), ("adapt_delta", C.c_double),
                ("stepsize", C.c_double), ("chains", C.c_int), ("chain_offset", C.c_int), ("metric", C.c_int)]


class NutsDiag(C.Structure):
    _fields_ = [("mean_e", C.c_double), ("min_e", C.c_double), ("max_e", C.c_double), ("divergent", C.c_longlong),
                ("treedepth_hits", C.c_longlong), ("batched_leapfrogs", C.c_longlong),
                ("stepsize_search_leapfrogs", C.c_longlong)]


def phase_ms(enable=True, reset=False):
    """host wall-clock per phase of the MCML iterations of this process since the last reset (csrc/trace.h)"""
    out = np.zeros(8)
    _lib.check(_lib.lib().glmmr_mcml_dbg_phase_ms(int(enable), int(reset), _p(out)))
    return {k: float(out[i]) for i, k in enumerate(("sample", "beta_step", "theta_step", "refresh"))}


def rccl_unique_id():
    """128 opaque bytes from ncclGetUniqueId: made on rank 0, handed to every rank's Context.comm_init_rccl"""
    buf = (C.c_ubyte * 128)()
    _lib.check(_lib.lib().glmmr_mcml_rccl_unique_id(buf))
    return bytes(buf)


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
        self._world = int(world)
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

    # -- native multi-GPU exchange (comm.hip): RCCL all-reduce on the context's stream
    def comm_init_rccl(self, unique_id, rank, world):
        buf = (C.c_ubyte * 128).from_buffer_copy(bytes(unique_id))
        _lib.check(_lib.lib().glmmr_mcml_ctx_comm_init_rccl(self._h, buf, int(rank), int(world)))
        self._world = int(world)

    def comm_allreduce(self, vals):
        """sum over the ranks through the path the statistics take (self-test of the exchange)"""
        v = np.ascontiguousarray(vals, dtype=np.float64).copy()
        _lib.check(_lib.lib().glmmr_mcml_ctx_comm_allreduce(self._h, _p(v), v.size))
        return v

    def comm_stats(self):
        calls = C.c_longlong(); dbl = C.c_longlong(); nat = C.c_int()
        _lib.check(_lib.lib().glmmr_mcml_ctx_comm_stats(self._h, C.byref(calls), C.byref(dbl), C.byref(nat)))
        return dict(calls=calls.value, doubles=dbl.value, native=bool(nat.value))

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

    def get_u_all(self, world=None):
        """every rank's samples, rank r's columns at r * mcols: the u mcml_full returns (mcml_full.cpp:144-145).
        Collective unless called right after mcml_full / mcml_optim (the theta-step has gathered them)."""
        w = int(world) if world else max(1, int(getattr(self, "_world", 1)))
        u = np.zeros((self.Q, self.mcols * w), order="F")
        nc = C.c_int()
        _lib.check(_lib.lib().glmmr_mcml_get_u_all(self._h, _p(u), self.Q, C.byref(nc)))
        assert nc.value == u.shape[1], (nc.value, u.shape)
        return u

    def shard_stats(self):
        out = (C.c_longlong * 6)()
        _lib.check(_lib.lib().glmmr_mcml_ctx_shard_stats(self._h, out))
        return dict(gathers=out[0], gather_doubles=out[1], theta_rounds=out[2], theta_evals_own=out[3],
                    theta_evals_all=out[4])

    def emulate_world(self, world, mode):
        """bench.py --as-rank-of N (include/glmmr_mcml_c.h glmmr_mcml_dbg_emulate_world): 1 record, 2 replay"""
        _lib.check(_lib.lib().glmmr_mcml_dbg_emulate_world(self._h, int(world), int(mode)))
        self._world = int(world) if world and world > 1 else 1

    # -- A8
    def mvn_ll(self, theta):
        theta = _f(theta).ravel()
        out = C.c_double()
        _lib.check(_lib.lib().glmmr_mcml_ctx_mvn_ll(self._h, _p(theta), C.byref(out)))
        return out.value

    def mvn_ll_batch(self, thetas):
        """mvn_ll at several thetas (rows of `thetas`) evaluated side by side on the device"""
        th = np.ascontiguousarray(np.atleast_2d(np.asarray(thetas, dtype=np.float64)))     # row j = candidate j = column-major R x k
        out = np.zeros(th.shape[0])
        _lib.check(_lib.lib().glmmr_mcml_ctx_mvn_ll_batch(self._h, _p(th), th.shape[0], _p(out)))
        return out

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

    def nuts_sample(self, beta, var_par, warmup, nsamp, seed, chains=1, chain_offset=0, iter_idx=0, max_treedepth=10,
                    adapt_delta=0.8, stepsize=1.0, want_trace=False, metric="diag_e"):
        """the sampler that stands where the reference calls Stan (gen_u_samples.R, inst/stan): csrc/nuts.h"""
        beta = _f(beta).ravel()
        o = NutsOpts(warmup, nsamp, max_treedepth, adapt_delta, stepsize, chains, chain_offset,
                     {"diag_e": 0, "unit_e": 1}[metric])
        d = NutsDiag()
        total = warmup + -(-nsamp // chains)
        c_ip_ = C.POINTER(C.c_int)
        tr = None
        if want_trace:
            tr = dict(depth=np.zeros((chains, total), dtype=np.int32, order="F"),
                      nleap=np.zeros((chains, total), dtype=np.int32, order="F"),
                      eps=np.zeros((chains, total), order="F"), accept=np.zeros((chains, total), order="F"))
        ncols = C.c_int()
        _lib.check(_lib.lib().glmmr_mcml_ctx_nuts_sample(
            self._h, _p(beta), C.c_double(var_par), C.byref(o), C.c_uint64(seed), C.c_uint32(iter_idx),
            None if tr is None else tr["depth"].ctypes.data_as(c_ip_), None if tr is None else tr["nleap"].ctypes.data_as(c_ip_),
            None if tr is None else _p(tr["eps"]), None if tr is None else _p(tr["accept"]), C.byref(d), C.byref(ncols)))
        self.mcols = ncols.value
        diag = dict(mean_e=d.mean_e, min_e=d.min_e, max_e=d.max_e, divergent=d.divergent, treedepth_hits=d.treedepth_hits,
                    batched_leapfrogs=d.batched_leapfrogs, stepsize_search_leapfrogs=d.stepsize_search_leapfrogs)
        return (diag, tr) if want_trace else diag

    # -- drivers on the resident context
    def _ext(self, seed, chains, maxfun, theta_batch=0):
        return Ext(int(seed or 0), int(chains or 1), int(maxfun or 0), 0, int(theta_batch or 0))

    def npar(self):
        return _lib.lib().glmmr_mcml_ctx_npar(self._h)

    def theta_log(self, enable=None, last=None):
        """test hook: the theta-step's objective evaluations, rows (theta..., log-likelihood); enable True clears + starts"""
        w = self.npar() + 1
        n = C.c_int()
        _lib.check(_lib.lib().glmmr_mcml_dbg_theta_log(self._h, -1, None, 0, C.byref(n)))
        rows = n.value if last is None else min(int(last), n.value)
        out = np.zeros((rows, w))
        _lib.check(_lib.lib().glmmr_mcml_dbg_theta_log(self._h, -1 if enable is None else int(bool(enable)),
                                                       _p(out) if rows else None, rows, C.byref(n)))
        return out

    def last_kernels(self):
        """kernel family of the sampler's last (forward, backward) product"""
        names = {-1: None, 0: "skinny", 1: "band", 2: "dlds", 3: "reg", 4: "sparse"}
        f = C.c_int(); b = C.c_int()
        _lib.check(_lib.lib().glmmr_mcml_ctx_last_kernels(self._h, C.byref(f), C.byref(b)))
        return names[f.value], names[b.value]

    def mcml_optim(self, start, trace=0, mcnr=False, maxfun=0, theta_batch=0):
        start = _f(start).ravel(); R = self.npar()
        b = np.zeros(self.P); t = np.zeros(R); sg = C.c_double()
        e = self._ext(0, 1, maxfun, theta_batch)
        _lib.check(_lib.lib().glmmr_mcml_ctx_optim(self._h, _p(start), start.size, int(trace), int(mcnr),
                                                   C.byref(e), _p(b), _p(t), C.byref(sg)))
        return dict(beta=b, theta=t, sigma=sg.value)

    def mcml_simlik(self, start, trace=0, maxfun=0, theta_batch=0):
        start = _f(start).ravel(); R = self.npar()
        b = np.zeros(self.P); t = np.zeros(R); sg = C.c_double()
        e = self._ext(0, 1, maxfun, theta_batch)
        _lib.check(_lib.lib().glmmr_mcml_ctx_simlik(self._h, _p(start), start.size, int(trace), C.byref(e),
                                                    _p(b), _p(t), C.byref(sg)))
        return dict(beta=b, theta=t, sigma=sg.value)

    def mcml_hess(self, start, tol=1e-5, trace=0):
        start = _f(start).ravel(); nv = self.P + self.npar()
        H = np.zeros((nv, nv), order="F")
        _lib.check(_lib.lib().glmmr_mcml_ctx_hess(self._h, _p(start), start.size, C.c_double(tol), int(trace), _p(H)))
        return H

    def aic_mcml(self, beta_par, cov_par):
        bp = _f(beta_par).ravel(); cp = _f(cov_par).ravel()
        out = C.c_double()
        _lib.check(_lib.lib().glmmr_mcml_ctx_aic(self._h, _p(bp), bp.size, _p(cp), cp.size, C.byref(out)))
        return out.value

    def mcml_full(self, start, mcnr=False, m=500, maxiter=30, warmup=500, tol=1e-3, verbose=False,
                  lambda_=0.05, trace=0, refresh=500, maxsteps=100, target_accept=0.9, seed=0, chains=1,
                  maxfun=0, theta_batch=0):
        start = _f(start).ravel(); R = self.npar()
        b = np.zeros(self.P); t = np.zeros(R); sg = C.c_double(); conv = C.c_int(); it = C.c_int()
        d = HmcDiag()
        e = self._ext(seed, chains, maxfun, theta_batch)
        _lib.check(_lib.lib().glmmr_mcml_ctx_full(
            self._h, _p(start), start.size, int(mcnr), int(m), int(maxiter), int(warmup), C.c_double(tol),
            int(verbose), C.c_double(lambda_), int(trace), int(refresh), int(maxsteps), C.c_double(target_accept),
            C.byref(e), _p(b), _p(t), C.byref(sg), C.byref(conv), C.byref(it), C.byref(d)))
        self.mcols = _lib.lib().glmmr_mcml_ctx_ncols(self._h)
        return dict(beta=b, theta=t, sigma=sg.value, converged=bool(conv.value), iters=it.value,
                    accept_rate=d.accept_rate, mean_e=d.mean_e, leapfrog_total=d.leapfrog_total)

    def mcml_la(self, start, usehess=False, tol=1e-3, verbose=False, trace=0, maxiter=10, nr=False, maxfun=0):
        """mcml_la / mcml_la_nr on the resident model (src/mcml_la.cpp:28-290)"""
        start = _f(start).ravel(); R = self.npar()
        b = np.zeros(self.P); t = np.zeros(R); sg = C.c_double(); conv = C.c_int(); it = C.c_int()
        se = np.zeros(start.size); u = np.zeros(self.Q)
        e = self._ext(0, 1, maxfun)
        _lib.check(_lib.lib().glmmr_mcml_ctx_la(
            self._h, _p(start), start.size, int(nr), int(usehess), C.c_double(tol), int(verbose), int(trace),
            int(maxiter), C.byref(e), _p(b), _p(t), C.byref(sg), _p(se), _p(u), C.byref(conv), C.byref(it)))
        return dict(beta=b, theta=t, sigma=sg.value, se=se, u=u, converged=bool(conv.value), iters=it.value)

    def la_probe(self, start, kind, v=None, var_par=1.0, par=None):
        """test hook: LA functor values (kind 0, 1, 2) or one mcnr_b step (kind 3)"""
        start = _f(start).ravel()
        vv = None if v is None else _f(v).ravel()
        if kind == 3:
            vo = np.zeros(self.Q); bo = np.zeros(self.P); so = C.c_double()
            _lib.check(_lib.lib().glmmr_mcml_dbg_la_probe(self._h, _p(start), start.size, 3, _p(vv),
                                                          C.c_double(var_par), None, 0, None, _p(vo), _p(bo),
                                                          C.byref(so)))
            return dict(v=vo, beta=bo, sigma=so.value)
        pr = _f(par).ravel(); out = C.c_double()
        _lib.check(_lib.lib().glmmr_mcml_dbg_la_probe(self._h, _p(start), start.size, int(kind), _p(vv),
                                                      C.c_double(var_par), _p(pr), pr.size, C.byref(out), None, None,
                                                      None))
        return out.value

    def profile(self, enable=True, reset=False):
        out = np.zeros(8)
        fa = C.c_longlong(); ba = C.c_longlong()
        _lib.check(_lib.lib().glmmr_mcml_ctx_profile_launches(self._h, C.byref(fa), C.byref(ba)))
        _lib.check(_lib.lib().glmmr_mcml_ctx_profile(self._h, int(enable), int(reset), _p(out)))
        # fwd_n / bwd_n: launches that were TIMED (the sampler times one proposal in four); *_all: every launch
        return dict(fwd_ms=out[0], fwd_n=int(out[1]), bwd_ms=out[2], bwd_n=int(out[3]), fwd_flops=out[4],
                    bwd_flops=out[5], dense_flops=out[6], operator=("dense", "banded", "sparse")[int(out[7])],
                    fwd_n_all=fa.value, bwd_n_all=ba.value)


def _problem(cov, data, eff_range, Z, X, y, family, link):
    keep = dict(cov=_i(cov), data=_f(data).ravel(), eff=_f(eff_range).ravel(), Z=_f(Z), X=_f(X), y=_f(y).ravel())
    p = Problem()
    p.cov = keep["cov"].ctypes.data_as(c_ip); p.cov_rows = keep["cov"].shape[0]
    p.data = _p(keep["data"]); p.data_len = keep["data"].size
    p.eff_range = _p(keep["eff"]); p.eff_len = keep["eff"].size
    p.Z = _p(keep["Z"]); p.X = _p(keep["X"]); p.y = _p(keep["y"])
    p.n, p.Q = keep["Z"].shape; p.P = keep["X"].shape[1]
    p.family = family.encode(); p.link = link.encode()
    return p, keep


# ---------------------------------------------------------------------------
# Mirrors of the Rcpp exports
# ---------------------------------------------------------------------------
def mcml_full(cov, data, eff_range, Z, X, y, family, link, start, mcnr=False, m=500, maxiter=30,
              warmup=500, tol=1e-3, verbose=True, lambda_=0.05, trace=0, refresh=500, maxsteps=100,
              target_accept=0.9, seed=0, chains=1, maxfun=0):
    """mcml_full(...) -> dict(beta, theta, sigma, converged, u)   (src/mcml_full.cpp:41-148).
    seed / chains / maxfun are the build's additions (reference: random_device, 1 chain, 10000)."""
    p, keep = _problem(cov, data, eff_range, Z, X, y, family, link)
    start = _f(start).ravel()
    L = _lib.lib()
    ncol = L.glmmr_mcml_sample_cols(int(m), int(chains))
    R = int(start.size - p.P - 1)
    b = np.zeros(p.P); t = np.zeros(R); sg = C.c_double(); conv = C.c_int(); uc = C.c_int()
    u = np.zeros((p.Q, ncol), order="F")
    e = Ext(int(seed), int(chains), int(maxfun), 0, 0)
    _lib.check(L.glmmr_mcml_full(C.byref(p), _p(start), start.size, int(mcnr), int(m), int(maxiter), int(warmup),
                                 C.c_double(tol), int(verbose), C.c_double(lambda_), int(trace), int(refresh),
                                 int(maxsteps), C.c_double(target_accept), C.byref(e), _p(b), _p(t), C.byref(sg),
                                 C.byref(conv), _p(u), p.Q, C.byref(uc)))
    return dict(beta=b, theta=t, sigma=sg.value, converged=bool(conv.value), u=u[:, :uc.value])


def _la_call(fn, cov, data, eff_range, Z, X, y, family, link, start, usehess, tol, verbose, trace, maxiter, maxfun):
    p, keep = _problem(cov, data, eff_range, Z, X, y, family, link)
    start = _f(start).ravel()
    R = int(start.size - p.P - 1)
    b = np.zeros(p.P); t = np.zeros(R); sg = C.c_double(); se = np.zeros(start.size); u = np.zeros(p.Q)
    e = Ext(0, 1, int(maxfun), 0, 0)
    _lib.check(fn(C.byref(p), _p(start), start.size, int(usehess), C.c_double(tol), int(verbose), int(trace),
                  int(maxiter), C.byref(e), _p(b), _p(t), C.byref(sg), _p(se), _p(u)))
    return dict(beta=b, theta=t, sigma=sg.value, se=se, u=u.reshape(-1, 1))


def mcml_la(cov, data, eff_range, Z, X, y, family, link, start, usehess=False, tol=1e-3, verbose=True, trace=0,
            maxiter=10, maxfun=0):
    """mcml_la(...) -> dict(beta, theta, sigma, se, u)   (src/mcml_la.cpp:28-155)"""
    return _la_call(_lib.lib().glmmr_mcml_la, cov, data, eff_range, Z, X, y, family, link, start, usehess, tol,
                    verbose, trace, maxiter, maxfun)


def mcml_la_nr(cov, data, eff_range, Z, X, y, family, link, start, usehess=False, tol=1e-3, verbose=True, trace=0,
               maxiter=10, maxfun=0):
    """mcml_la_nr(...) -> dict(beta, theta, sigma, se, u)   (src/mcml_la.cpp:174-290)"""
    return _la_call(_lib.lib().glmmr_mcml_la_nr, cov, data, eff_range, Z, X, y, family, link, start, usehess, tol,
                    verbose, trace, maxiter, maxfun)


def mcmc_sample(Z, L, X, y, beta, family, link, warmup, nsamp, lambda_, var_par=1, trace=0, refresh=500,
                maxsteps=100, target_accept=0.9, seed=0, chains=1):
    """mcmc_sample(...) -> Q x (nsamp+1) matrix of u = L v   (src/mcml_full.cpp:314-338)"""
    Z = _f(Z); Lm = _f(L); X = _f(X); y = _f(y).ravel(); beta = _f(beta).ravel()
    n, Q = Z.shape
    lib = _lib.lib()
    ncol = lib.glmmr_mcml_sample_cols(int(nsamp), int(chains))
    out = np.zeros((Q, ncol), order="F"); nc = C.c_int()
    e = Ext(int(seed), int(chains), 0, 0, 0)
    _lib.check(lib.glmmr_mcml_mcmc_sample(_p(Z), _p(Lm), _p(X), _p(y), n, Q, X.shape[1], _p(beta),
                                          family.encode(), link.encode(), int(warmup), int(nsamp),
                                          C.c_double(lambda_), C.c_double(var_par), int(trace), int(refresh),
                                          int(maxsteps), C.c_double(target_accept), C.byref(e), _p(out), Q,
                                          C.byref(nc)))
    return out[:, :nc.value]


def gen_u_samples(y, X, Z, L, beta, family, link, sigma=1.0, warmup_iter=100, m=100, seed=0, chains=1,
                  max_treedepth=10, adapt_delta=0.8):
    """gen_u_samples(y, X, Z, L, beta, family, sigma, warmup_iter, m) -> Q x m matrix of u = L gamma
    (R/gen_u_samples.R:38-69; the NUTS sampler of csrc/nuts.h stands where cmdstanr is called)"""
    Z = _f(Z); Lm = _f(L); X = _f(X); y = _f(y).ravel(); beta = _f(beta).ravel()
    n, Q = Z.shape
    ncol = chains * -(-int(m) // chains)
    out = np.zeros((Q, ncol), order="F"); nc = C.c_int()
    e = Ext(int(seed), int(chains), 0, 0, 0)
    o = NutsOpts(int(warmup_iter), int(m), int(max_treedepth), float(adapt_delta), 0.0, int(chains), 0, 0)
    _lib.check(_lib.lib().glmmr_mcml_gen_u_samples(_p(Z), _p(Lm), _p(X), _p(y), n, Q, X.shape[1], _p(beta), family.encode(),
                                                   link.encode(), C.c_double(sigma), int(warmup_iter), int(m), C.byref(o),
                                                   C.byref(e), _p(out), Q, C.byref(nc)))
    return out[:, :nc.value]


def _fit_call(fn, prob_args, u, start, extra, sparse=None, maxfun=0):
    p, keep = _problem(*prob_args)
    u = _f(u); start = _f(start).ravel()
    R = int(start.size - p.P - 1) if start.size > p.P + 1 else 0
    b = np.zeros(p.P); t = np.zeros(max(R, 64)); sg = C.c_double()
    e = Ext(0, 1, int(maxfun), 0, 0)
    args = [C.byref(p)]
    if sparse is not None:
        Ap, Ai = _i(sparse[0]).ravel(), _i(sparse[1]).ravel()
        args += [Ap.ctypes.data_as(c_ip), Ai.ctypes.data_as(c_ip), Ai.size]
    args += [_p(u), u.shape[1], _p(start), start.size] + extra + [C.byref(e), _p(b), _p(t), C.byref(sg)]
    _lib.check(fn(*args))
    return b, t, sg.value


def _npar_of(cov):
    fnpar = [0, 1, 1, 1, 2, 2, 1, 2, 2, 2, 2, 2, 2, 2, 1]
    cov = _i(cov)
    return int(max(cov[r, 4] + fnpar[cov[r, 2]] for r in range(cov.shape[0])))


def mcml_optim(cov, data, eff_range, Z, X, y, u, family, link, start, trace=0, mcnr=False, maxfun=0):
    """mcml_optim(...) -> dict(beta, theta, sigma)   (src/mcml_optim.cpp:35-68)"""
    b, t, s = _fit_call(_lib.lib().glmmr_mcml_optim, (cov, data, eff_range, Z, X, y, family, link), u, start,
                        [int(trace), int(mcnr)], maxfun=maxfun)
    return dict(beta=b, theta=t[:_npar_of(cov)], sigma=s)


def mcml_simlik(cov, data, eff_range, Z, X, y, u, family, link, start, trace=0, maxfun=0):
    """mcml_simlik(...)   (src/mcml_optim.cpp:90-117)"""
    b, t, s = _fit_call(_lib.lib().glmmr_mcml_simlik, (cov, data, eff_range, Z, X, y, family, link), u, start,
                        [int(trace)], maxfun=maxfun)
    return dict(beta=b, theta=t[:_npar_of(cov)], sigma=s)


def mcml_optim_sparse(cov, data, eff_range, Ap, Ai, Z, X, y, u, family, link, start, trace=0, mcnr=False,
                      maxfun=0):
    """mcml_optim_sparse(...) -> dict(beta, theta, sigma, Ap, Ai, Ax, D)   (src/mcml_optim.cpp:147-184):
    Ap/Ai/Ax = the unit lower LDL' factor of D(theta) without its diagonal, D its pivots."""
    p, keep = _problem(cov, data, eff_range, Z, X, y, family, link)
    u = _f(u); start = _f(start).ravel()
    cv = keep["cov"]
    lib = _lib.lib()
    cap = lib.glmmr_mcml_sparse_factor_nnz(cv.ctypes.data_as(c_ip), cv.shape[0], _p(keep["data"]), keep["data"].size)
    R = _npar_of(cov)
    b = np.zeros(p.P); t = np.zeros(R); sg = C.c_double()
    Lp = np.zeros(p.Q + 1, dtype=np.int32); Li = np.zeros(max(cap, 1), dtype=np.int32)
    Lx = np.zeros(max(cap, 1)); D = np.zeros(p.Q)
    Ap_, Ai_ = _i(Ap).ravel(), _i(Ai).ravel()
    e = Ext(0, 1, int(maxfun), 0, 0)
    _lib.check(lib.glmmr_mcml_optim_sparse(C.byref(p), Ap_.ctypes.data_as(c_ip), Ai_.ctypes.data_as(c_ip), Ai_.size,
                                           _p(u), u.shape[1], _p(start), start.size, int(trace), int(mcnr),
                                           C.byref(e), _p(b), _p(t), C.byref(sg), Lp.ctypes.data_as(c_ip),
                                           Li.ctypes.data_as(c_ip), _p(Lx), _p(D), cap))
    return dict(beta=b, theta=t, sigma=sg.value, Ap=Lp, Ai=Li[:cap], Ax=Lx[:cap], D=D)


def mcml_simlik_sparse(cov, data, eff_range, Ap, Ai, Z, X, y, u, family, link, start, trace=0, maxfun=0):
    """mcml_simlik_sparse(...)   (src/mcml_optim.cpp:210-239)"""
    b, t, s = _fit_call(_lib.lib().glmmr_mcml_simlik_sparse, (cov, data, eff_range, Z, X, y, family, link), u,
                        start, [int(trace)], sparse=(Ap, Ai), maxfun=maxfun)
    return dict(beta=b, theta=t[:_npar_of(cov)], sigma=s)


def mcml_hess(cov, data, eff_range, Z, X, y, u, family, link, start, tol=1e-5, trace=0, sparse=None):
    """mcml_hess(...) -> (P+R) x (P+R)   (src/mcml_optim.cpp:263-285; _sparse :313-337)"""
    p, keep = _problem(cov, data, eff_range, Z, X, y, family, link)
    u = _f(u); start = _f(start).ravel()
    nv = p.P + _npar_of(cov)
    H = np.zeros((nv, nv), order="F")
    e = Ext(0, 1, 0, 0, 0)
    if sparse is None:
        _lib.check(_lib.lib().glmmr_mcml_hess(C.byref(p), _p(u), u.shape[1], _p(start), start.size,
                                              C.c_double(tol), int(trace), C.byref(e), _p(H)))
    else:
        Ap, Ai = _i(sparse[0]).ravel(), _i(sparse[1]).ravel()
        _lib.check(_lib.lib().glmmr_mcml_hess_sparse(C.byref(p), Ap.ctypes.data_as(c_ip), Ai.ctypes.data_as(c_ip),
                                                     Ai.size, _p(u), u.shape[1], _p(start), start.size,
                                                     C.c_double(tol), int(trace), C.byref(e), _p(H)))
    return H


def mcml_hess_sparse(cov, data, eff_range, Ap, Ai, Z, X, y, u, family, link, start, tol=1e-5, trace=0):
    return mcml_hess(cov, data, eff_range, Z, X, y, u, family, link, start, tol, trace, sparse=(Ap, Ai))


def aic_mcml(cov, data, eff_range, Z, X, y, u, family, link, beta_par, cov_par):
    """aic_mcml(...) -> double   (src/mcml_optim.cpp:356-392)"""
    p, keep = _problem(cov, data, eff_range, Z, X, y, family, link)
    u = _f(u); bp = _f(beta_par).ravel(); cp = _f(cov_par).ravel()
    out = C.c_double(); e = Ext(0, 1, 0, 0, 0)
    _lib.check(_lib.lib().glmmr_mcml_aic(C.byref(p), _p(u), u.shape[1], _p(bp), bp.size, _p(cp), cp.size,
                                         C.byref(e), C.byref(out)))
    return out.value



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
