"""Host optimiser (BOBYQA family) and R-style finite differences: pure host code of
the product library, exercised through its C ABI on the CPU.

rminqa (the reference's optimiser, mcmloptim.h:56-113) is not in the image, so
its trajectory cannot be pinned; what is checked is that the optimiser finds
the known optimum of standard bound-constrained problems, and that the finite
differences follow R's optimhess (exact on quadratics, SURVEY 8c KAT 7)."""
import ctypes as C

import numpy as np
import pytest
from scipy import optimize

dp = C.POINTER(C.c_double)
OBJ = C.CFUNCTYPE(C.c_double, dp, C.c_int, C.c_void_p)


def _bobyqa(fun, x0, lower=None, upper=None, rhobeg=0.0, rhoend=0.0, maxfun=0):
    from glmmrmcml_amd import _lib
    L = _lib.lib()
    n = len(x0)
    calls = []

    def cb(xp, nn, user):
        x = np.array([xp[i] for i in range(nn)])
        calls.append(x)
        return float(fun(x))
    cbk = OBJ(cb)
    x0 = np.asarray(x0, float); out = np.zeros(n); f = C.c_double(); nf = C.c_int()
    lo = None if lower is None else np.asarray(lower, float)
    up = None if upper is None else np.asarray(upper, float)
    rc = L.glmmr_mcml_dbg_bobyqa(cbk, None, n, x0.ctypes.data_as(dp),
                                 None if lo is None else lo.ctypes.data_as(dp),
                                 None if up is None else up.ctypes.data_as(dp),
                                 C.c_double(rhobeg), C.c_double(rhoend), maxfun,
                                 out.ctypes.data_as(dp), C.byref(f), C.byref(nf))
    _lib.check(rc)
    return out, f.value, nf.value, calls


def test_quadratic_unconstrained():
    rng = np.random.default_rng(0)
    for n in (2, 3, 6, 11):
        A = rng.normal(size=(n, n)); A = A @ A.T + n * np.eye(n)
        xs = rng.normal(size=n)
        x, f, nf, _ = _bobyqa(lambda x: 0.5 * (x - xs) @ A @ (x - xs) + 3.0, xs + 1.0)
        assert np.abs(x - xs).max() < 2e-6 and abs(f - 3.0) < 1e-10
        assert nf < 60 * n


def test_rosenbrock_with_and_without_bounds():
    ros = lambda x: 100 * (x[1] - x[0] ** 2) ** 2 + (1 - x[0]) ** 2
    x, f, nf, _ = _bobyqa(ros, [-1.2, 1.0], rhoend=1e-8)
    assert np.abs(x - 1).max() < 1e-5 and nf < 600
    # active bound: x0 <= 0.5 -> optimum at (0.5, 0.25)
    x, f, nf, calls = _bobyqa(ros, [-1.2, 1.0], lower=[-2, -2], upper=[0.5, 2], rhoend=1e-8)
    assert abs(x[0] - 0.5) < 1e-7 and abs(x[1] - 0.25) < 1e-5
    allx = np.array(calls)
    assert allx[:, 0].max() <= 0.5 and allx.min() >= -2      # never evaluates outside the bounds


def test_matches_scipy_on_bounded_problems():
    rng = np.random.default_rng(3)
    for n in (2, 4, 7):
        A = rng.normal(size=(n, n)); A = A @ A.T + np.eye(n)
        b = rng.normal(size=n) * 3
        fun = lambda x: 0.5 * x @ A @ x - b @ x + np.sum(np.exp(0.3 * x))
        lo = np.full(n, 1e-6); up = np.full(n, np.inf)
        x, f, nf, _ = _bobyqa(fun, np.full(n, 0.5), lower=lo, upper=up, rhoend=1e-9)
        ref = optimize.minimize(fun, np.full(n, 0.5), bounds=[(1e-6, None)] * n, method="L-BFGS-B",
                                options=dict(ftol=1e-15, gtol=1e-12))
        assert f <= ref.fun + 1e-9 * max(1, abs(ref.fun))
        assert np.abs(x - ref.x).max() < 1e-4


def test_one_parameter_and_flat_direction():
    x, f, nf, _ = _bobyqa(lambda x: (x[0] - 0.3) ** 2, [1.0], lower=[1e-6], upper=[np.inf])
    assert abs(x[0] - 0.3) < 1e-6
    # a parameter the objective ignores (sigma in f_optim, mcmloptim.h:102-105) does not disturb the others
    x, f, nf, _ = _bobyqa(lambda x: (x[0] - 2) ** 2 + (x[1] + 1) ** 2, [0.0, 0.0, 0.7], lower=[-9, -9, 0.0],
                          upper=[9, 9, 9])
    assert np.abs(x[:2] - [2, -1]).max() < 1e-5 and 0.0 <= x[2] <= 9


def test_default_settings_follow_minqa():
    # rhobeg = min(0.95, 0.2 max|x0|): the first coordinate step is x0 + rhobeg e_1
    _, _, _, calls = _bobyqa(lambda x: float(np.sum(x ** 2)), [2.0, -1.0])
    assert np.allclose(calls[0], [2.0, -1.0]) and np.allclose(calls[1], [2.4, -1.0])
    _, _, nf, _ = _bobyqa(lambda x: float(np.sum(x ** 2)), [2.0, -1.0], maxfun=7)
    assert nf <= 8


def test_fd_hessian_is_optimhess():
    from glmmrmcml_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(5)
    n = 4
    A = rng.normal(size=(n, n)); A = A + A.T
    g = rng.normal(size=n)
    fun = lambda x: 0.5 * x @ A @ x + g @ x
    cbk = OBJ(lambda xp, nn, u: float(fun(np.array([xp[i] for i in range(nn)]))))
    x = rng.normal(size=n); H = np.zeros((n, n))
    _lib.check(L.glmmr_mcml_dbg_fd_hessian(cbk, None, n, x.ctypes.data_as(dp), C.c_double(1e-3), 0, None, None,
                                           H.ctypes.data_as(dp)))
    assert np.abs(H - A).max() < 1e-7          # exact on a quadratic up to rounding
    assert np.array_equal(H, H.T)
    # one-sided at a bound (usebounds = 1)
    lo = x.copy(); up = x + 10
    f3 = lambda x: float(np.sum(x ** 3))
    cb3 = OBJ(lambda xp, nn, u: f3(np.array([xp[i] for i in range(nn)])))
    _lib.check(L.glmmr_mcml_dbg_fd_hessian(cb3, None, n, x.ctypes.data_as(dp), C.c_double(1e-4), 1,
                                           lo.ctypes.data_as(dp), up.ctypes.data_as(dp), H.ctypes.data_as(dp)))
    assert np.all(np.isfinite(H))


# ---------------------------------------------------------------- the batch schedule (csrc/optim.h bobyqa_batch)
# What the theta-step of a chain-sharded job runs (drivers.hip d_optim_sharded): `width` candidates per round, one per
# rank.  No reference counterpart (the reference is one process); the contract is the sequential optimiser's: the same
# optimum of the same objective -- in far fewer ROUNDS than the sequential run needs evaluations.
def _bobyqa_batch(fun, x0, width, lower=None, upper=None, rhobeg=0.0, rhoend=0.0, maxfun=0):
    from glmmrmcml_amd import _lib
    L = _lib.lib()
    n = len(x0)
    calls = []

    def cb(xp, nn, user):
        x = np.array([xp[i] for i in range(nn)])
        calls.append(x)
        return float(fun(x))
    cbk = OBJ(cb)
    x0 = np.asarray(x0, float); out = np.zeros(n); f = C.c_double(); nf = C.c_int(); rd = C.c_int()
    lo = None if lower is None else np.asarray(lower, float)
    up = None if upper is None else np.asarray(upper, float)
    rc = L.glmmr_mcml_dbg_bobyqa_batch(cbk, None, n, x0.ctypes.data_as(dp),
                                       None if lo is None else lo.ctypes.data_as(dp),
                                       None if up is None else up.ctypes.data_as(dp),
                                       C.c_double(rhobeg), C.c_double(rhoend), maxfun, width,
                                       out.ctypes.data_as(dp), C.byref(f), C.byref(nf), C.byref(rd))
    _lib.check(rc)
    return out, f.value, nf.value, rd.value, calls


def _mvn_objective(Q=120, m=24, seed=5):
    """-(1/m) sum_j log N(u_j; 0, D(theta)), D = theta_0 exp(-d / theta_1): the theta-step's objective
    (mcmldmatrix.h:23-41) in numpy, samples drawn at theta = (0.25, 0.1)"""
    import scipy.linalg as sla
    rng = np.random.default_rng(seed)
    xy = rng.uniform(size=(Q, 2))
    dist = np.sqrt(((xy[:, None, :] - xy[None, :, :]) ** 2).sum(-1))
    U = np.linalg.cholesky(0.25 * np.exp(-dist / 0.1)) @ rng.standard_normal((Q, m))

    def f(th):
        try:
            Lc = np.linalg.cholesky(th[0] * np.exp(-dist / th[1]))
        except np.linalg.LinAlgError:
            return 1e300
        z = sla.solve_triangular(Lc, U, lower=True)
        return -(-0.5 * Q * np.log(2 * np.pi) - np.log(np.diag(Lc)).sum() - 0.5 * (z * z).sum() / m)
    return f


@pytest.mark.parametrize("width", [1, 2, 3, 4, 8, 16])
def test_batch_schedule_finds_the_sequential_optimum(width):
    rng = np.random.default_rng(0)
    for n in (1, 2, 3, 5):
        A = rng.normal(size=(n, n)); A = A @ A.T + n * np.eye(n)
        xs = rng.normal(size=n)
        x, f, nf, rounds, _ = _bobyqa_batch(lambda x: 0.5 * (x - xs) @ A @ (x - xs) + 3.0, xs + 1.0, width)
        assert np.abs(x - xs).max() < 2e-6 and abs(f - 3.0) < 1e-10, (n, width)
        assert rounds * width >= nf > 0 and (width == 1 or rounds < nf)
    ros = lambda x: 100 * (x[1] - x[0] ** 2) ** 2 + (1 - x[0]) ** 2
    x, f, nf, rounds, calls = _bobyqa_batch(ros, [-1.2, 1.0], width, lower=[-2, -2], upper=[0.5, 2], rhoend=1e-8)
    assert abs(x[0] - 0.5) < 1e-7 and abs(x[1] - 0.25) < 1e-5
    allx = np.array(calls)
    assert allx[:, 0].max() <= 0.5 and allx.min() >= -2          # never evaluates outside the bounds
    x, f, nf, rounds, _ = _bobyqa_batch(lambda x: (x[0] - 0.3) ** 2, [1.0], width, lower=[1e-6], upper=[np.inf])
    assert abs(x[0] - 0.3) < 1e-6


def test_batch_schedule_on_the_theta_step_objective():
    """the MVN objective over log(theta) as d_optim_sharded runs it (rhobeg 0.25, rhoend 1e-7): same optimum as the
    sequential optimiser over theta, a fraction of its sequential depth, identical on every rank (deterministic)"""
    f = _mvn_objective()
    g = lambda z: f(np.exp(z))
    xs, fs, nfs, _ = _bobyqa(f, [0.4, 0.15], lower=[1e-6] * 2, upper=[np.inf] * 2)
    for start in ([0.4, 0.15], [0.25, 0.1], [0.1, 0.3]):
        for width in (2, 8):
            z, fb, nf, rounds, calls = _bobyqa_batch(g, np.log(start), width, lower=[np.log(1e-6)] * 2,
                                                     upper=[np.inf] * 2, rhobeg=0.25, rhoend=1e-7)
            assert abs(fb - fs) < 1e-9 * abs(fs) and np.abs(np.exp(z) - xs).max() < 2e-6 * np.abs(xs).max(), (start, width)
            if width == 8:
                assert rounds <= 20 and rounds < nfs / 3, (rounds, nfs)        # sequential depth: 10-16 vs ~60-100
            z2, fb2, nf2, rounds2, calls2 = _bobyqa_batch(g, np.log(start), width, lower=[np.log(1e-6)] * 2,
                                                          upper=[np.inf] * 2, rhobeg=0.25, rhoend=1e-7)
            assert np.array_equal(z, z2) and nf == nf2 and all(np.array_equal(a, b) for a, b in zip(calls, calls2))


def test_batch_schedule_under_a_fixed_budget():
    """bench.py's theta-step budget (40 evaluations): eight candidates per round = five or six rounds, and the
    objective reached is no worse than what the sequential optimiser reaches with the same 40 evaluations"""
    f = _mvn_objective()
    g = lambda z: f(np.exp(z))
    fopt = _bobyqa(f, [0.25, 0.1], lower=[1e-6] * 2, upper=[np.inf] * 2)[1]
    for start in ([0.25, 0.1], [0.3, 0.12], [0.2, 0.08]):
        _, fseq, nfs, _ = _bobyqa(f, start, lower=[1e-6] * 2, upper=[np.inf] * 2, maxfun=40)
        z, fb, nf, rounds, _ = _bobyqa_batch(g, np.log(start), 8, lower=[np.log(1e-6)] * 2, upper=[np.inf] * 2,
                                             rhobeg=0.25, rhoend=1e-7, maxfun=40)
        assert nf <= 40 and rounds <= 7
        assert fb - fopt <= max(fseq - fopt, 1e-6), (start, fb - fopt, fseq - fopt)


def test_batch_objective_failure_and_nan():
    # NaN objective values are "+inf" as in the sequential optimiser; maxfun inside the initial design still answers
    x, f, nf, rounds, _ = _bobyqa_batch(lambda x: np.nan if x[0] < 0 else (x[0] - 1) ** 2 + x[1] ** 2, [2.0, 1.0], 4)
    assert np.abs(x - [1, 0]).max() < 1e-5
    x, f, nf, rounds, _ = _bobyqa_batch(lambda x: float(np.sum(x ** 2)), [2.0, -1.0], 4, maxfun=3)
    assert nf <= 4 and np.isfinite(f)


def test_batch_schedule_survives_points_without_a_value():
    """a candidate at which the objective has no value (+inf / NaN: D(theta) not positive definite there) is 'worse than
    anything seen', not an infinity inside the interpolation model: the run still converges to the sequential optimum"""
    f = _mvn_objective()

    def g(z):
        th = np.exp(z)
        return np.inf if th[1] > 0.12 else f(th)         # the first step along theta_2 of the initial design fails
    ref = _bobyqa(f, [0.25, 0.1], lower=[1e-6] * 2, upper=[np.inf] * 2)
    assert ref[0][1] < 0.12
    for width in (4, 8):
        z, fb, nf, rounds, _ = _bobyqa_batch(g, np.log([0.25, 0.1]), width, lower=[np.log(1e-6)] * 2, upper=[np.inf] * 2,
                                             rhobeg=0.25, rhoend=1e-7)
        assert abs(fb - ref[1]) < 1e-9 * abs(ref[1]) and np.abs(np.exp(z) - ref[0]).max() < 2e-6, (width, np.exp(z), ref[0])
