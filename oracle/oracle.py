"""ctypes front end of oracle/libmcml_oracle.so (TEST INFRASTRUCTURE ONLY).

PARITY UNPINNED at the glmmrBase / SparseChol / rminqa boundary: see
oracle/mcml_oracle.h.  Everything is column-major float64 / int32, as R hands
it to the reference's Rcpp exports (src/RcppExports.cpp:15-310).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

# OpenMP must not oversubscribe the CPUs this process may actually use (a GPU box
# exposes many hardware threads but grants a small share): bound the team and
# make idle threads sleep instead of spin.  Set before libgomp initialises.
try:
    _NCPU = len(os.sched_getaffinity(0))
except AttributeError:                      # pragma: no cover
    _NCPU = os.cpu_count() or 1
os.environ.setdefault("OMP_NUM_THREADS", str(max(1, min(_NCPU, 16))))
os.environ.setdefault("OMP_WAIT_POLICY", "passive")
os.environ.setdefault("OMP_DYNAMIC", "false")

c_dp = C.POINTER(C.c_double)
c_ip = C.POINTER(C.c_int32)
c_u8p = C.POINTER(C.c_uint8)


class HmcOpts(C.Structure):
    _fields_ = [("warmup", C.c_int), ("nsamp", C.c_int), ("adapt", C.c_int),
                ("lambda_", C.c_double), ("max_steps", C.c_int),
                ("target_accept", C.c_double)]


class HmcDiag(C.Structure):
    _fields_ = [("accept", C.c_int), ("e", C.c_double), ("ebar", C.c_double),
                ("steps", C.c_int)]


def build(force=False):
    so = os.path.join(_HERE, "libmcml_oracle.so")
    src = os.path.join(_HERE, "mcml_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libmcml_oracle.so")
        if not os.path.exists(so):
            so = build()
        L = C.CDLL(so)
        L.orc_u52.restype = C.c_double
        L.orc_dlog.restype = C.c_double
        L.orc_dlog.argtypes = [C.c_double]
        L.orc_ppnd16.restype = C.c_double
        L.orc_ppnd16.argtypes = [C.c_double]
        L.orc_normal.restype = C.c_double
        L.orc_normal.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
        L.orc_minstd_canonical.restype = C.c_double
        L.orc_minstd_next.restype = C.c_uint32
        L.orc_chain_minstd_seed.restype = C.c_uint32
        L.orc_chain_minstd_seed.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32]
        L.orc_log_factorial_approx.restype = C.c_double
        L.orc_log_factorial_approx.argtypes = [C.c_double]
        L.orc_logpdf.restype = C.c_double
        L.orc_logpdf.argtypes = [C.c_double, C.c_double, C.c_double, C.c_int]
        for f in (L.orc_mod_inv, L.orc_dhdmu, L.orc_detadmu):
            f.restype = C.c_double
            f.argtypes = [C.c_double, C.c_int]
        L.orc_digamma.restype = C.c_double
        L.orc_digamma.argtypes = [C.c_double]
        L.orc_log_prob.restype = C.c_double
        L.orc_log_prob.argtypes = [C.c_int, C.c_int, c_dp, c_dp, c_dp, C.c_double, C.c_int, c_dp]
        L.orc_log_grad.argtypes = [C.c_int, C.c_int, c_dp, c_dp, c_dp, C.c_double, C.c_int, c_dp, c_dp]
        L.orc_model_loglik.restype = C.c_double
        L.orc_model_loglik.argtypes = [C.c_int, C.c_int, C.c_int, c_dp, c_dp, c_dp, c_dp, C.c_int,
                                       C.c_double, C.c_int, C.c_int, c_dp]
        L.orc_mvn_ll.argtypes = [c_ip, C.c_int, c_dp, c_dp, c_dp, c_dp, C.c_int, C.c_int, C.c_int, c_dp]
        L.orc_gen_D.argtypes = [c_ip, C.c_int, c_dp, c_dp, c_dp, C.c_int, c_dp]
        L.orc_hmc_chain.argtypes = [C.c_int, C.c_int, c_dp, c_dp, c_dp, C.c_double, C.c_int,
                                    C.POINTER(HmcOpts), C.c_uint64, C.c_uint32, C.c_uint32,
                                    c_dp, c_dp, c_dp, c_u8p, c_dp, C.POINTER(HmcDiag)]
        L.orc_mcnr.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, c_dp, c_dp, c_dp, c_dp, C.c_int,
                               c_dp, C.c_double, C.c_int, C.c_int, C.c_int,
                               c_dp, c_dp, c_dp, c_dp, c_dp]
        L.orc_gemm_nn.argtypes = [C.c_int, C.c_int, C.c_int, c_dp, C.c_int, c_dp, C.c_int, c_dp, C.c_int]
        # libgomp may have been initialised (by torch / numpy) before the environment variables
        # above were set: bound the team explicitly as well
        L.orc_set_num_threads(int(os.environ.get("OMP_NUM_THREADS", "16")))
        _LIB = L
    return _LIB


def num_threads():
    return lib().orc_num_threads()


def _d(a):
    return None if a is None else a.ctypes.data_as(c_dp)


def _f(a):
    """float64, Fortran (column-major) contiguous"""
    return np.asfortranarray(np.asarray(a, dtype=np.float64))


def _cov(cov):
    cov = np.asfortranarray(np.asarray(cov, dtype=np.int32))
    assert cov.ndim == 2 and cov.shape[1] == 5
    return cov


# ---- RNG -------------------------------------------------------------------
def philox(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().orc_philox4x32_10(c, k, o)
    return [int(x) for x in o]


def normal(seed, elem, chain, prop, tag):
    return lib().orc_normal(seed, elem, chain, prop, tag)


def minstd_canonical_stream(seed, n):
    x = C.c_uint32(seed % 2147483647 or 1)
    return [lib().orc_minstd_canonical(C.byref(x)) for _ in range(n)]


# ---- GLM --------------------------------------------------------------------
def flink(family, link):
    return lib().orc_flink(family.encode(), link.encode())


def link_code(link):
    return lib().orc_link_code(link.encode())


def logpdf(y, mu, var_par, fl):
    return lib().orc_logpdf(float(y), float(mu), float(var_par), int(fl))


# ---- covariance -------------------------------------------------------------
def cov_npar(cov):
    cov = _cov(cov)
    return lib().orc_cov_npar(cov.ctypes.data_as(c_ip), cov.shape[0])


def cov_N(cov):
    cov = _cov(cov)
    return lib().orc_cov_N(cov.ctypes.data_as(c_ip), cov.shape[0])


def gen_D(cov, data, eff_range, gamma, chol=False):
    cov = _cov(cov)
    data = _f(data); eff = _f(eff_range); gamma = _f(gamma)
    N = cov_N(cov)
    D = np.zeros((N, N), order="F")
    rc = lib().orc_gen_D(cov.ctypes.data_as(c_ip), cov.shape[0], _d(data), _d(eff), _d(gamma),
                         int(chol), _d(D))
    if rc:
        raise RuntimeError("orc_gen_D failed rc=%d" % rc)
    return D


def mvn_ll(cov, data, eff_range, gamma, u, per_column_refactor=False):
    """export mvn_ll (src/mcml_optim.cpp:406-414)"""
    cov = _cov(cov)
    data = _f(data); eff = _f(eff_range); gamma = _f(gamma); u = _f(u)
    if u.ndim == 1:
        u = u.reshape(-1, 1, order="F")
    out = C.c_double()
    rc = lib().orc_mvn_ll(cov.ctypes.data_as(c_ip), cov.shape[0], _d(data), _d(eff), _d(gamma),
                          _d(u), u.shape[0], u.shape[1], int(per_column_refactor), C.byref(out))
    if rc:
        raise RuntimeError("orc_mvn_ll failed rc=%d" % rc)
    return out.value


# ---- model ------------------------------------------------------------------
def digamma(x):
    return lib().orc_digamma(float(x))


def dhdmu(eta, fl):
    """glmmrBase maths::dhdmu as restated in mcml_oracle.c (the ONE table: csrc/glm.h and model.py follow it)"""
    return np.array([lib().orc_dhdmu(float(e), int(fl)) for e in np.atleast_1d(eta)])


def log_prob(xb, ZL, y, var_par, fl, v):
    xb = _f(xb); ZL = _f(ZL); y = _f(y); v = _f(v)
    n, Q = ZL.shape
    return lib().orc_log_prob(n, Q, _d(xb), _d(ZL), _d(y), float(var_par), int(fl), _d(v))


def log_grad(xb, ZL, y, var_par, fl, v):
    xb = _f(xb); ZL = _f(ZL); y = _f(y); v = _f(v)
    n, Q = ZL.shape
    g = np.zeros(Q)
    lib().orc_log_grad(n, Q, _d(xb), _d(ZL), _d(y), float(var_par), int(fl), _d(v), _d(g))
    return g


def model_loglik(Z, xb, y, u, var_par, fl, ncols=None):
    """mcmlModel::log_likelihood (mcmlmodel.h:284-304); first `ncols` columns"""
    Z = _f(Z); xb = _f(xb); y = _f(y); u = _f(u)
    n, Q = Z.shape
    m = u.shape[1] if ncols is None else ncols
    return lib().orc_model_loglik(n, Q, m, _d(Z), _d(xb), _d(y), _d(u), u.shape[0],
                                  float(var_par), int(fl), 1, None)


def hmc_chain(xb, ZL, y, var_par, fl, warmup, nsamp, lambda_, max_steps, target_accept,
              seed, chain_id=0, iter_idx=0, adapt=100, inj_init=None, inj_mom=None):
    """one chain of mcmcRunHMC::sample (mhmcmc.h:121-157); returns whitened
    samples Q x (nsamp+1), accept flags, probs, diag"""
    xb = _f(xb); ZL = _f(ZL); y = _f(y)
    n, Q = ZL.shape
    o = HmcOpts(warmup, nsamp, adapt, lambda_, max_steps, target_accept)
    samples = np.zeros((Q, nsamp + 1), order="F")
    flags = np.zeros(warmup + nsamp, dtype=np.uint8)
    probs = np.zeros(warmup + nsamp)
    diag = HmcDiag()
    ii = None if inj_init is None else _f(inj_init)
    im = None if inj_mom is None else _f(inj_mom)
    rc = lib().orc_hmc_chain(n, Q, _d(xb), _d(ZL), _d(y), float(var_par), int(fl), C.byref(o),
                             int(seed), int(chain_id), int(iter_idx), _d(ii), _d(im),
                             _d(samples), flags.ctypes.data_as(c_u8p), _d(probs), C.byref(diag))
    if rc:
        raise RuntimeError("orc_hmc_chain rc=%d" % rc)
    return samples, flags, probs, dict(accept=diag.accept, e=diag.e, ebar=diag.ebar, steps=diag.steps)


def mcnr(X, Z, y, u, beta, var_par, family, link, ncols=None):
    """mcmloptim::mcnr (mcmloptim.h:198-236) -> dict(beta, sigma, XtWX, XtWr, sigma_sum)"""
    X = _f(X); Z = _f(Z); y = _f(y); u = _f(u); beta = _f(beta)
    n, P = X.shape
    Q = Z.shape[1]
    m = u.shape[1] if ncols is None else ncols
    fl = flink(family, link)
    S1 = np.zeros((P, P), order="F"); S2 = np.zeros(P); S3 = C.c_double()
    bout = np.zeros(P); sout = C.c_double()
    rc = lib().orc_mcnr(n, Q, P, m, _d(X), _d(Z), _d(y), _d(u), u.shape[0], _d(beta),
                        float(var_par), fl, link_code(link), 0,
                        _d(S1), _d(S2), C.byref(S3), _d(bout), C.byref(sout))
    if rc:
        raise RuntimeError("orc_mcnr rc=%d" % rc)
    return dict(beta=bout, sigma=sout.value, XtWX=S1, XtWr=S2, sigma_sum=S3.value)


def gemm(A, B):
    A = _f(A); B = _f(B)
    M, K = A.shape
    N = B.shape[1]
    Cm = np.zeros((M, N), order="F")
    lib().orc_gemm_nn(M, N, K, _d(A), M, _d(B), K, _d(Cm), M)
    return Cm
