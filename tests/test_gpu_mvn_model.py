"""GPU parity: mvn_ll (A8), genD, log_likelihood (A6), mcnr (A7) through the C
ABI vs the CPU oracle on the same seeded inputs.  Tolerance: f64 with a
different summation order -> 1e-10 relative (north star asks 1e-6)."""
import numpy as np
import pytest

from glmmrmcml_amd import synth

pytestmark = pytest.mark.gpu
RTOL = 1e-10


def _rel(a, b):
    return abs(a - b) / max(1e-300, abs(b))


@pytest.mark.parametrize("Q,m", [(8, 3), (33, 5), (128, 7), (129, 64), (300, 17), (641, 33), (1500, 96)])
def test_mvn_ll_dense_block(orc, Q, m):
    from glmmrmcml_amd import api
    d = synth.geospatial(Q, seed=Q)
    rng = np.random.default_rng(Q + 1)
    u = rng.normal(size=(Q, m)) * 0.5
    want = orc.mvn_ll(d["cov"], d["data"], d["eff_range"], d["theta"], u)
    got = api.mvn_ll(d["cov"], d["data"], d["eff_range"], d["theta"], u)
    assert _rel(got, want) < RTOL
    # a second theta on the same context (what BOBYQA does)
    with api.Context(d["cov"], d["data"], d["eff_range"]) as ctx:
        ctx.set_u(u)
        for th in ([0.4, 0.2], [0.1, 0.05]):
            assert _rel(ctx.mvn_ll(th), orc.mvn_ll(d["cov"], d["data"], d["eff_range"], th, u)) < RTOL


def test_mvn_ll_diagonal_blocks(orc):
    from glmmrmcml_amd import api
    d = synth.cluster_rct(ncl=10, nt=5, nind=2, seed=1)
    rng = np.random.default_rng(2)
    u = rng.normal(size=(d["Q"], 101)) * 0.2
    want = orc.mvn_ll(d["cov"], d["data"], d["eff_range"], d["theta"], u)
    got = api.mvn_ll(d["cov"], d["data"], d["eff_range"], d["theta"], u)
    assert _rel(got, want) < RTOL
    # closed form (SURVEY 8c KAT 1)
    th = d["theta"]
    dd = np.r_[np.full(10, th[0] ** 2), np.full(50, th[1] ** 2)][:, None]
    cf = np.mean((-0.5 * np.log(2 * np.pi * dd) - 0.5 * u ** 2 / dd).sum(0))
    assert _rel(got, cf) < 1e-10


def test_mvn_ll_small_ar1_blocks(orc):
    from glmmrmcml_amd import api
    d = synth.stepped_wedge(ncl=40, nt=8, nind=1, seed=4)
    rng = np.random.default_rng(5)
    u = rng.normal(size=(d["Q"], 130)) * 0.3
    want = orc.mvn_ll(d["cov"], d["data"], d["eff_range"], d["theta"], u)
    got = api.mvn_ll(d["cov"], d["data"], d["eff_range"], d["theta"], u)
    assert _rel(got, want) < RTOL


def test_mvn_ll_single_column_and_errors(orc):
    from glmmrmcml_amd import api, _lib
    d = synth.geospatial(40, seed=3)
    u = np.random.default_rng(1).normal(size=40)
    assert _rel(api.mvn_ll(d["cov"], d["data"], d["eff_range"], d["theta"], u),
                orc.mvn_ll(d["cov"], d["data"], d["eff_range"], d["theta"], u)) < RTOL
    # not positive definite: negative variance parameter
    with pytest.raises(_lib.McmlError) as e:
        api.mvn_ll(d["cov"], d["data"], d["eff_range"], [-0.25, 0.1], u)
    assert e.value.code == -3
    # unknown covariance function id
    bad = d["cov"].copy(); bad[0, 2] = 5
    with pytest.raises(_lib.McmlError) as e:
        api.mvn_ll(bad, d["data"], d["eff_range"], d["theta"], u)
    assert e.value.code == -2
    # too few parameters / wrong Q
    with pytest.raises(_lib.McmlError):
        api.mvn_ll(d["cov"], d["data"], d["eff_range"], [0.25], u)
    with pytest.raises(_lib.McmlError):
        api.mvn_ll(d["cov"], d["data"], d["eff_range"], d["theta"], u[:-1])


@pytest.mark.parametrize("gen,kw", [(synth.geospatial, dict(n=200)), (synth.geospatial, dict(n=517)),
                                    (synth.stepped_wedge, dict(ncl=6, nt=5, nind=2)),
                                    (synth.cluster_rct, dict(ncl=4, nt=3, nind=2))])
def test_gen_D_and_chol(orc, gen, kw):
    from glmmrmcml_amd import api
    d = gen(**kw)
    with api.Context(d["cov"], d["data"], d["eff_range"]) as ctx:
        D = ctx.gen_D(d["theta"], chol=False)
        L = ctx.gen_D(d["theta"], chol=True)
    Do = orc.gen_D(d["cov"], d["data"], d["eff_range"], d["theta"])
    Lo = orc.gen_D(d["cov"], d["data"], d["eff_range"], d["theta"], chol=True)
    assert np.allclose(D, Do, rtol=1e-13, atol=0)
    # entries of L span many orders of magnitude: compare in the norm, and by reconstruction
    assert np.abs(L - Lo).max() < 1e-10 * np.abs(Lo).max()
    assert np.abs(L @ L.T - Do).max() < 1e-13 * max(1.0, np.abs(Do).max()) * D.shape[0]
    assert np.allclose(np.triu(L, 1), 0)


@pytest.mark.parametrize("gen,kw,vp", [(synth.geospatial, dict(n=150), 0.8),
                                       (synth.cluster_rct, dict(ncl=6, nt=4, nind=5), 1.0),
                                       (synth.cluster_rct, dict(ncl=6, nt=4, nind=5, family="poisson"), 1.0),
                                       (synth.stepped_wedge, dict(ncl=7, nt=4, nind=6), 1.0)])
def test_loglik_and_mcnr(orc, gen, kw, vp):
    from glmmrmcml_amd import api
    d = gen(**kw)
    rng = np.random.default_rng(11)
    m = 37
    u = rng.normal(size=(d["Q"], m + 1)) * 0.3
    beta = d["beta"] + 0.1 * rng.normal(size=d["P"])
    fl = orc.flink(d["family"], d["link"])
    with api.Context(d["cov"], d["data"], d["eff_range"], d["Z"], d["X"], d["y"], d["family"], d["link"]) as ctx:
        ctx.set_u(u, niter=m)          # D5 quirk: beta-step reads m columns, theta-step m+1
        got = ctx.loglik(beta, vp)
        want = orc.model_loglik(d["Z"], d["X"] @ beta, d["y"], u, vp, fl, ncols=m)
        assert _rel(got, want) < RTOL
        assert _rel(ctx.mvn_ll(d["theta"]), orc.mvn_ll(d["cov"], d["data"], d["eff_range"], d["theta"], u)) < RTOL
        r = ctx.mcnr(beta, vp)
        ro = orc.mcnr(d["X"], d["Z"], d["y"], u, beta, vp, d["family"], d["link"], ncols=m)
        assert np.allclose(r["XtWX"], ro["XtWX"], rtol=1e-11)
        assert np.allclose(r["XtWr"], ro["XtWr"], rtol=1e-9, atol=1e-9)
        assert np.allclose(r["beta"], ro["beta"], rtol=1e-9, atol=1e-11)
        assert _rel(r["sigma"], ro["sigma"]) < 1e-11


def test_unknown_family_link_is_an_error():
    from glmmrmcml_amd import api, _lib
    d = synth.cluster_rct(ncl=3, nt=2, nind=2)
    with pytest.raises(_lib.McmlError) as e:
        api.Context(d["cov"], d["data"], d["eff_range"], d["Z"], d["X"], d["y"], "gaussian", "cloglog")
    assert e.value.code == -2


def test_mvn_ll_batch_equals_single_evaluations(orc):
    """glmmr_mcml_ctx_mvn_ll_batch: k candidate thetas factorised side by side give the values of k single evaluations
    (the very bits below the size at which the batch regroups its trailing updates into K = 1024 passes, 1e-12 above it),
    the same bits on every repeat (eager, captured, replayed), a non-positive-definite candidate comes back as NaN
    without disturbing its neighbours, and the values agree with the oracle (mcmldmatrix.h:23-41)"""
    from glmmrmcml_amd import api
    for d, m in ((synth.geospatial(300, seed=2), 40), (synth.geospatial(1500, seed=2), 64),
                 (synth.stepped_wedge(ncl=6, nt=4, nind=5), 16), (synth.cluster_rct(ncl=6, nt=4, nind=5), 12)):
        u = np.asfortranarray(np.random.default_rng(4).standard_normal((d["Q"], m)))
        with api.Context(d["cov"], d["data"], d["eff_range"]) as ctx:
            ctx.set_u(u)
            T = np.array([np.asarray(d["theta"]) * (1 + 0.05 * k) for k in range(5)])
            single = np.array([ctx.mvn_ll(t) for t in T])
            first = None
            for rep in range(3):                       # eager, captured, replayed
                got = ctx.mvn_ll_batch(T)
                if d["Q"] <= 1024 + 128:
                    assert np.array_equal(got, single), (rep, got, single)
                else:
                    assert np.allclose(got, single, rtol=1e-12, atol=0), (rep, got, single)
                first = got if first is None else first
                assert np.array_equal(got, first)
            assert np.array_equal(ctx.mvn_ll_batch(T[:1]), single[:1])
            if d["Q"] <= 400:
                want = np.array([orc.mvn_ll(d["cov"], d["data"], d["eff_range"], t, u) for t in T])
                assert np.abs(got - want).max() < 1e-10 * np.abs(want).max()
    # a candidate outside the positive-definite region (an AR1 parameter of 1.5) is NaN; the others are untouched
    d = synth.stepped_wedge(ncl=6, nt=4, nind=5)
    u = np.asfortranarray(np.random.default_rng(4).standard_normal((d["Q"], 8)))
    with api.Context(d["cov"], d["data"], d["eff_range"]) as ctx:
        ctx.set_u(u)
        good = np.asarray(d["theta"], float); bad = good.copy(); bad[1] = 1.5
        got = ctx.mvn_ll_batch(np.array([good, bad, good * 1.1]))
        assert np.isnan(got[1]) and got[0] == ctx.mvn_ll(good) and got[2] == ctx.mvn_ll(good * 1.1)


def test_mvn_ll_batch_odd_shapes():
    """block sizes round multiples of the 128-wide panel and of the 1024-wide super-panel (where a batch regroups its trailing
    updates), one or a few sample columns, 2 / 3 / 8 candidates: every batched value equals the single evaluation to rounding"""
    from glmmrmcml_amd import api
    for n in (33, 129, 257, 1023, 1153, 1296, 2049):
        d = synth.geospatial(n, seed=n)
        for m in (1, 3, 17):
            u = np.asfortranarray(np.random.default_rng(m).standard_normal((n, m)))
            with api.Context(d["cov"], d["data"], d["eff_range"]) as ctx:
                ctx.set_u(u)
                for k in (2, 3, 8):
                    T = np.array([np.asarray(d["theta"]) * (1 + 0.04 * j) for j in range(k)])
                    single = np.array([ctx.mvn_ll(t) for t in T])
                    for rep in range(3):
                        got = ctx.mvn_ll_batch(T)
                        assert np.abs(got - single).max() < 1e-11 * np.abs(single).max(), (n, m, k, rep, got, single)
