"""GPU parity of the No-U-Turn sampler (csrc/nuts.h) -- the sampler that stands where the reference calls Stan
(R/gen_u_samples.R:38-69, inst/stan/mcml_*.stan) -- against oracle/nuts.py.

PARITY UNPINNED: cmdstan does not exist in this image and the reference holds no Stan output.  The oracle restates the
published algorithm with the device's random streams; every transition's tree depth and leapfrog count must be
identical, step sizes within 1e-8 relative and draws within 1e-6 for the short runs, 1e-5 / 1e-3 after 24 adaptive
transitions (f64, different summation orders; no decision in these seeded cases sits within rounding of its threshold).  The exact gaussian posterior is the known answer."""
import numpy as np
import pytest

from glmmrmcml_amd import synth

pytestmark = pytest.mark.gpu


def _setup(orc, d, api):
    ctx = api.Context(d["cov"], d["data"], d["eff_range"], d["Z"], d["X"], d["y"], d["family"], d["link"])
    ctx.update_L(d["theta"])
    Lo = orc.gen_D(d["cov"], d["data"], d["eff_range"], d["theta"], chol=True)
    return ctx, d["Z"] @ Lo, d["X"] @ d["beta"], orc.flink(d["family"], d["link"]), Lo


CASES = [(synth.geospatial, dict(n=96), 0.9),                                  # dense ZL: column-major state, MFMA products
         (synth.cluster_rct, dict(ncl=6, nt=4, nind=5, family="poisson"), 1.0),  # sparse ZL as a product: chain-major state
         (synth.stepped_wedge, dict(ncl=6, nt=4, nind=30), 1.0)]               # sparse ZL as two factors


# warm-up 14: fewer than 20 transitions, Stan adapts the step size only; 24: init_buffer 3, one window of 19
# transitions whose end (transition 21) replaces the metric, searches the step size again and restarts the dual
# averaging, term_buffer 2; unit_e: the same 24 transitions without metric adaptation
@pytest.mark.parametrize("warm,metric", [(14, "diag_e"), (24, "diag_e"), (24, "unit_e")])
@pytest.mark.parametrize("gen,kw,vp", CASES)
def test_chains_match_oracle_transition_by_transition(orc, gen, kw, vp, warm, metric):
    from glmmrmcml_amd import api
    from oracle import nuts
    d = gen(**kw)
    ctx, ZL, xb, fl, Lo = _setup(orc, d, api)
    Cn, nsamp, seed, it, md = 4, 12, 20240607, 3, 6
    diag, tr = ctx.nuts_sample(d["beta"], vp, warm, nsamp, seed, chains=Cn, chain_offset=5, iter_idx=it, max_treedepth=md,
                               want_trace=True, metric=metric)
    u = ctx.get_u()
    dpc = 3
    assert u.shape == (d["Q"], Cn * dpc)
    ndiv = nhit = 0
    for c in range(Cn):
        so, to, dg = nuts.nuts_chain(xb, ZL, d["y"], vp, fl, warm, dpc, seed, chain_id=5 + c, iter_idx=it, max_treedepth=md,
                                     metric=metric)
        assert np.array_equal(tr["depth"][c], to["depth"]), (c, tr["depth"][c], to["depth"])
        assert np.array_equal(tr["nleap"][c], to["nleap"]), (c, tr["nleap"][c], to["nleap"])
        # rounding differences of the acceptance statistic go through the dual averaging (gain sqrt(t) / 0.05 on log eps)
        # and the trajectories amplify them (a factor ~10 every few transitions on the poisson case): 1e-9 after 14
        # transitions, up to 2e-6 after 24 observed -- the integer decisions (depth, leapfrog count) stay identical
        tol = 1e-8 if warm < 20 else 1e-5
        assert np.abs(tr["eps"][c] / to["eps"] - 1).max() < tol
        assert np.abs(tr["accept"][c] - to["accept"]).max() < tol
        uo = Lo @ so
        assert np.abs(u[:, c * dpc:(c + 1) * dpc] - uo).max() < 100 * tol * max(1.0, np.abs(uo).max())
        ndiv += to["ndiv"]; nhit += to["nhit"]
    assert diag["divergent"] == ndiv and diag["treedepth_hits"] == nhit
    assert tr["depth"].max() >= 2                       # the trees did grow
    ctx.close()


def test_many_chains_recover_gaussian_posterior(orc):
    """gaussian-identity: the posterior of gamma is N(mu*, S*) exactly (SURVEY 8c KAT 5)"""
    from glmmrmcml_amd import api
    d = synth.geospatial(40, seed=21)
    ctx, ZL, xb, fl, Lo = _setup(orc, d, api)
    S = np.linalg.inv(np.eye(40) + ZL.T @ ZL / d["sigma"] ** 2)
    mu = S @ ZL.T @ (d["y"] - xb) / d["sigma"] ** 2
    diag = ctx.nuts_sample(d["beta"], d["sigma"], 60, 1024, seed=5, chains=1024)
    u = ctx.get_u()
    assert u.shape == (40, 1024)
    v = np.linalg.solve(Lo, u)
    se = np.sqrt(np.diag(S) / 1024)
    assert np.all(np.abs(v.mean(1) - mu) < 5 * se)
    assert np.abs(np.cov(v) - S).max() < 0.12 * np.abs(S).max()
    # dual averaging opens, and reopens after the metric update, near 10 x the searched step size: early divergences
    assert diag["divergent"] < 0.06 * 1024 * 61 and 1e-3 < diag["mean_e"] < 10
    ctx.close()


def test_gen_u_samples_export(orc):
    """gen_u_samples(y, X, Z, L, beta, family, sigma, warmup_iter, m) (R/gen_u_samples.R:38-69): Q x m draws of
    u = L gamma, the same draws as the context-level call with the same seed"""
    from glmmrmcml_amd import api
    d = synth.cluster_rct(ncl=6, nt=3, nind=6, seed=4)
    L = orc.gen_D(d["cov"], d["data"], d["eff_range"], d["theta"], chol=True)
    with api.Context(d["cov"], d["data"], d["eff_range"], d["Z"], d["X"], d["y"], d["family"], d["link"]) as ctx:
        ctx.set_L(L)                                    # the export takes L as a matrix: the dense operator
        ctx.nuts_sample(d["beta"], 1.0, 20, 16, seed=99, chains=1)
        want = ctx.get_u()
    got = api.gen_u_samples(d["y"], d["X"], d["Z"], L, d["beta"], d["family"], d["link"], 1.0, warmup_iter=20, m=16, seed=99)
    assert got.shape == (d["Q"], 16) and np.isfinite(got).all()
    assert np.abs(got - want).max() < 1e-9
    assert np.abs(got).max() > 0


def test_sharded_chains_equal_unsharded_chains():
    """a chain's draws depend on its GLOBAL id only (momenta, initial state and the uniform stream are keyed by it):
    ranks that own chains [0, 2) and [2, 4) reproduce the 4-chain run column for column -- and the packing of the
    chains that still grow (csrc/nuts.h) does not depend on who else is in the batch"""
    from glmmrmcml_amd import api
    d = synth.geospatial(60, seed=8)
    out = {}
    with api.Context(d["cov"], d["data"], d["eff_range"], d["Z"], d["X"], d["y"], d["family"], d["link"]) as ctx:
        ctx.update_L(d["theta"])
        for name, (C, off) in dict(all=(4, 0), lo=(2, 0), hi=(2, 2)).items():
            dg, tr = ctx.nuts_sample(d["beta"], d["sigma"], 12, 2 * C, seed=31, chains=C, chain_offset=off, iter_idx=1,
                                     max_treedepth=7, want_trace=True)
            out[name] = (ctx.get_u(), tr)
    u, tr = out["all"]
    assert np.array_equal(tr["depth"][:2], out["lo"][1]["depth"]) and np.array_equal(tr["depth"][2:], out["hi"][1]["depth"])
    assert np.array_equal(tr["nleap"][:2], out["lo"][1]["nleap"]) and np.array_equal(tr["nleap"][2:], out["hi"][1]["nleap"])
    # 2 draws per chain, chain-major columns; the products of a 2-chain run are the streamed kernels, of the 4-chain
    # run the MFMA tiles: equal to rounding, not bit for bit
    assert np.abs(u[:, :4] - out["lo"][0]).max() < 1e-8 * max(1.0, np.abs(u).max())
    assert np.abs(u[:, 4:] - out["hi"][0]).max() < 1e-8 * max(1.0, np.abs(u).max())


def test_full_size_run_hits_the_gaussian_posterior():
    """config 2's size (n = Q = 2000, 256 chains): 40 adaptive transitions with every mechanism in play -- metric
    window, re-searched step size, packing of the chains that still grow down to the streamed products -- then the
    closed-form posterior mean of u = L gamma as in the fixed-length sampler's full-size test"""
    from glmmrmcml_amd import api
    n, chains = 2000, 256
    d = synth.geospatial(n)
    with api.Context(d["cov"], d["data"], d["eff_range"], d["Z"], d["X"], d["y"], d["family"], d["link"]) as ctx:
        ctx.update_L(d["theta"])
        diag, tr = ctx.nuts_sample(d["beta"], d["sigma"], 40, chains, seed=3, chains=chains, want_trace=True)
        u = ctx.get_u()
        L = ctx.gen_D(d["theta"], chol=True)
    assert u.shape == (n, chains) and np.isfinite(u).all()
    assert tr["depth"][:, -1].min() >= 1 and tr["depth"].max() <= 10 and (tr["nleap"] <= 1023).all()
    assert 0.5 < tr["accept"][:, 30:].mean() <= 1.0
    s2 = d["sigma"] ** 2
    S = np.linalg.inv(np.eye(n) + L.T @ L / s2)
    mu_u = L @ (S @ (L.T @ (d["y"] - d["X"] @ d["beta"]))) / s2
    sd_u = np.sqrt(np.einsum("ij,jk,ik->i", L, S, L) / chains)
    err = u.mean(1) - mu_u
    assert np.mean(np.abs(err) < 6 * sd_u + 0.1 * np.abs(mu_u).max()) > 0.99
