"""GPU parity of the sampler (A3-A5) vs the CPU oracle.

RNG: bit-exact (integer Philox/minstd streams, exactly-rounded arithmetic).
log_prob / log_grad: 1e-11 relative (f64, different summation order).
Chains: accept/reject decisions identical, samples within 1e-8: HMC on a
log-concave target contracts rounding differences, and no draw in these seeded
cases sits within 1e-9 of its acceptance threshold (asserted)."""
import ctypes as C

import numpy as np
import pytest

from glmmrmcml_amd import synth

pytestmark = pytest.mark.gpu
dp = C.POINTER(C.c_double)


def test_device_rng_is_bit_identical_to_oracle(orc):
    from glmmrmcml_amd import _lib
    L = _lib.lib()
    n = 5000
    out = np.zeros(n)
    _lib.check(L.glmmr_mcml_dbg_normals(C.c_uint64(0x123456789abcdef), 7, 11, 34, n, out.ctypes.data_as(dp)))
    want = np.array([orc.normal(0x123456789abcdef, i, 7, 11, 34) for i in range(n)])
    assert np.array_equal(out, want)
    assert abs(out.mean()) < 0.05 and abs(out.std() - 1) < 0.05
    u = np.zeros(256)
    for seed in (1, 12345, 2147483646):
        _lib.check(L.glmmr_mcml_dbg_minstd(seed, 256, u.ctypes.data_as(dp)))
        assert np.array_equal(u, np.array(orc.minstd_canonical_stream(seed, 256)))


def _setup(orc, d, api):
    ctx = api.Context(d["cov"], d["data"], d["eff_range"], d["Z"], d["X"], d["y"], d["family"], d["link"])
    ctx.update_L(d["theta"])
    Lo = orc.gen_D(d["cov"], d["data"], d["eff_range"], d["theta"], chol=True)
    return ctx, d["Z"] @ Lo, d["X"] @ d["beta"], orc.flink(d["family"], d["link"]), Lo


CASES = [(synth.geospatial, dict(n=96), 0.9), (synth.geospatial, dict(n=333), 1.0),
         (synth.cluster_rct, dict(ncl=6, nt=4, nind=5), 1.0),
         (synth.cluster_rct, dict(ncl=6, nt=4, nind=5, family="poisson"), 1.0),
         (synth.stepped_wedge, dict(ncl=9, nt=4, nind=5), 1.0),
         # nnz(ZL) / Q = 75 >= 24: the wave-per-random-effect backward kernel config 4 selects at full size
         (synth.stepped_wedge, dict(ncl=6, nt=4, nind=30), 1.0)]


@pytest.fixture(params=["skinny", "mfma"])
def few_chain_path(request, monkeypatch):
    """at most 16 chains run the streamed products of dgemm_skinny.h unless GLMMR_MCML_SKINNY=0: the few-chain parity
    tests run with both settings, so the MFMA kernels and their fused epilogues keep their chain-by-chain check"""
    if request.param == "mfma":
        monkeypatch.setenv("GLMMR_MCML_SKINNY", "0")
    else:
        monkeypatch.delenv("GLMMR_MCML_SKINNY", raising=False)
    return request.param


@pytest.mark.parametrize("gen,kw,vp", CASES)
def test_log_prob_and_log_grad(orc, gen, kw, vp, few_chain_path):
    from glmmrmcml_amd import api
    d = gen(**kw)
    ctx, ZL, xb, fl, _ = _setup(orc, d, api)
    rng = np.random.default_rng(3)
    ncol = 3 if few_chain_path == "skinny" else 5          # 3: the 4-wide instantiation of the streamed products
    V = rng.normal(size=(d["Q"], ncol)) * 0.7
    lp, G = ctx.log_prob_grad(d["beta"], vp, V)
    for c in range(ncol):
        lo = orc.log_prob(xb, ZL, d["y"], vp, fl, V[:, c])
        go = orc.log_grad(xb, ZL, d["y"], vp, fl, V[:, c])
        assert abs(lp[c] - lo) < 1e-11 * abs(lo)
        assert np.abs(G[:, c] - go).max() < 1e-11 * max(1.0, np.abs(go).max())
    ctx.close()


@pytest.mark.parametrize("gen,kw,vp", CASES[:1] + CASES[2:])
def test_chains_match_oracle_chain_by_chain(orc, gen, kw, vp, few_chain_path):
    from glmmrmcml_amd import api
    d = gen(**kw)
    ctx, ZL, xb, fl, Lo = _setup(orc, d, api)
    Cn, warm, nsamp, lam, ms, ta, seed, it = 5, 14, 15, 0.4, 6, 0.9, 20240601, 2
    if few_chain_path == "skinny":
        Cn = 3                                                 # the 4-wide instantiation of the streamed products
    diag, flags, probs = ctx.hmc_sample(d["beta"], vp, warm, nsamp, lam, ms, ta, seed, chains=Cn,
                                        chain_offset=10, iter_idx=it, adapt=10, want_trace=True)
    u = ctx.get_u()
    dpc = -(-nsamp // Cn)           # draws per chain
    assert u.shape == (d["Q"], Cn * dpc)
    acc_total = 0
    for c in range(Cn):
        so, fo, po, dg = orc.hmc_chain(xb, ZL, d["y"], vp, fl, warm, dpc, lam, ms, ta, seed,
                                       chain_id=10 + c, iter_idx=it, adapt=10)
        assert np.array_equal(flags[c], fo), (c, flags[c], fo)
        assert np.abs(probs[c] - po).max() < 1e-9
        uo = Lo @ so[:, 1:]
        assert np.abs(u[:, c * dpc:(c + 1) * dpc] - uo).max() < 1e-8 * max(1.0, np.abs(uo).max())
        acc_total += dg["accept"]
    assert abs(diag["accept_rate"] - acc_total / (Cn * (warm + dpc))) < 1e-12
    ctx.close()


def test_single_chain_reference_layout_and_injection(orc, few_chain_path):
    """chains=1 reproduces the reference's Q x (nsamp+1) output (mhmcmc.h:126,142,155) and the
    niter quirk (D5); injected initial state / momenta give the same chain as the generated ones"""
    from glmmrmcml_amd import api
    d = synth.cluster_rct(ncl=5, nt=3, nind=4, seed=12)
    ctx, ZL, xb, fl, Lo = _setup(orc, d, api)
    Q = d["Q"]
    warm, nsamp, seed = 8, 9, 77
    ctx.hmc_sample(d["beta"], 1.0, warm, nsamp, 0.3, 5, 0.9, seed)
    u = ctx.get_u()
    assert u.shape == (Q, nsamp + 1)
    so, fo, po, _ = orc.hmc_chain(xb, ZL, d["y"], 1.0, fl, warm, nsamp, 0.3, 5, 0.9, seed)
    assert np.abs(u - Lo @ so).max() < 1e-8
    init = np.array([orc.normal(seed, k, 0, 0, 0) for k in range(Q)])
    mom = np.array([[orc.normal(seed, k, 0, it, 2) for k in range(Q)] for it in range(warm + nsamp)]).T
    diag, fl2, pr2 = ctx.hmc_sample(d["beta"], 1.0, warm, nsamp, 0.3, 5, 0.9, seed, inj_init=init,
                                    inj_mom=mom, want_trace=True)
    assert np.array_equal(ctx.get_u(), u) and np.array_equal(fl2[0], fo)
    # beta-step reads nsamp columns, theta-step nsamp+1
    ll = ctx.loglik(d["beta"], 1.0)
    assert abs(ll - orc.model_loglik(d["Z"], xb, d["y"], u, 1.0, fl, ncols=nsamp)) < 1e-10 * abs(ll)
    ctx.close()


def test_many_chains_recover_gaussian_posterior(orc):
    """gaussian-identity posterior of v is N(mu*, S*) exactly (SURVEY 8c KAT 5)"""
    from glmmrmcml_amd import api
    d = synth.geospatial(40, seed=21)
    ctx, ZL, xb, fl, Lo = _setup(orc, d, api)
    S = np.linalg.inv(np.eye(40) + ZL.T @ ZL / d["sigma"] ** 2)
    mu = S @ ZL.T @ (d["y"] - xb) / d["sigma"] ** 2
    diag = ctx.hmc_sample(d["beta"], d["sigma"], 150, 2048, 1.5, 20, 0.9, seed=5, chains=2048)
    u = ctx.get_u()
    assert u.shape == (40, 2048)
    v = np.linalg.solve(Lo, u)
    se = np.sqrt(np.diag(S) / 2048)
    assert np.all(np.abs(v.mean(1) - mu) < 5 * se)
    assert np.abs(np.cov(v) - S).max() < 0.12 * np.abs(S).max()
    assert 0.6 < diag["accept_rate"] <= 1.0
    ctx.close()


def test_sparse_and_dense_zl_operators_agree(monkeypatch):
    """configs 1/4/5: the ELL/CSR ZL operator (as the product ZL and as the two factors Z, L applied in turn,
    hmc_cm.h) and the dense n x Q products on the FP64 MFMA kernels (20 chains: more than the streamed few-column
    kernel takes; asserted) are the same sampler"""
    from glmmrmcml_amd import api
    for gen, kw in ((synth.stepped_wedge, dict(ncl=9, nt=6, nind=12)), (synth.longitudinal, dict(nsubj=50, nvisit=4)),
                    (synth.stepped_wedge, dict(ncl=7, nt=5, nind=40)),       # long rows of ZL' (config 4's regime)
                    (synth.cluster_rct, dict(ncl=6, nt=4, nind=5, family="poisson"))):
        d = gen(**kw)
        out = {}
        for mode in ("product", "factored", "dense"):
            monkeypatch.setenv("GLMMR_MCML_ZL", mode)
            with api.Context(d["cov"], d["data"], d["eff_range"], d["Z"], d["X"], d["y"], d["family"], d["link"]) as ctx:
                ctx.update_L(d["theta"])
                diag, flags, probs = ctx.hmc_sample(d["beta"], 1.0, 15, 40, 0.4, 8, 0.9, seed=77, chains=20,
                                                    want_trace=True)
                kinds = set(ctx.last_kernels())
                assert kinds == {"sparse"} if mode != "dense" else kinds <= {"band", "dlds", "reg"}, (mode, kinds)
                V = np.random.default_rng(5).normal(size=(d["Q"], 3)) * 0.5
                lp, G = ctx.log_prob_grad(d["beta"], 1.0, V)
                out[mode] = (ctx.get_u(), flags.copy(), probs.copy(), lp, G)
        monkeypatch.delenv("GLMMR_MCML_ZL", raising=False)
        for mode in ("factored", "dense"):
            assert np.array_equal(out["product"][1], out[mode][1]), mode
            assert np.abs(out["product"][2] - out[mode][2]).max() < 1e-9, mode
            assert np.abs(out["product"][0] - out[mode][0]).max() < 1e-8, mode
            assert np.abs(out["product"][3] - out[mode][3]).max() < 1e-11 * np.abs(out["product"][3]).max(), mode
            assert np.abs(out["product"][4] - out[mode][4]).max() < 1e-11 * max(1.0, np.abs(out["product"][4]).max()), mode


def test_fused_block_kernel_of_the_factored_operator_is_bit_identical(monkeypatch):
    """factored sparse operator: k_cm_Lcol of a leapfrog step and k_cm_Lrow of the next one as ONE launch (a wave per covariance
    block, hmc_cm.h::k_cm_Lcol_Lrow) against the two kernels (GLMMR_MCML_CM_LFUSE=0): the same draws bit for bit -- blocks of 5 and
    8 (the 8-wide instantiation), 12 (the 16-wide one), step counts below and at the cap, a ragged last chain group"""
    from glmmrmcml_amd import api
    monkeypatch.setenv("GLMMR_MCML_ZL", "factored")
    for kw, lam, ms, chains in ((dict(ncl=7, nt=5, nind=40), 0.4, 10, 70), (dict(ncl=6, nt=8, nind=30), 5.0, 6, 64),
                                (dict(ncl=5, nt=12, nind=25), 0.5, 8, 130)):
        d = synth.stepped_wedge(**kw)
        out = []
        with api.Context(d["cov"], d["data"], d["eff_range"], d["Z"], d["X"], d["y"], d["family"], d["link"]) as ctx:
            ctx.update_L(d["theta"])
            for fuse in ("1", "0"):
                monkeypatch.setenv("GLMMR_MCML_CM_LFUSE", fuse)
                diag, flags, probs = ctx.hmc_sample(d["beta"], 1.0, 25, chains, lam, ms, 0.9, seed=11, chains=chains, want_trace=True)
                assert set(ctx.last_kernels()) == {"sparse"}
                out.append((ctx.get_u(), flags.copy(), probs.copy(), diag))
        assert np.array_equal(out[0][1], out[1][1]), kw
        assert np.array_equal(out[0][2], out[1][2]), kw
        assert np.array_equal(out[0][0], out[1][0]), kw
        assert out[0][3]["leapfrog_total"] == out[1][3]["leapfrog_total"]
    monkeypatch.delenv("GLMMR_MCML_CM_LFUSE", raising=False)


def test_speculative_launch_of_the_step_cap_changes_nothing(tmp_path):
    """while the longest chain sits at the step cap the sampler launches the cap without waiting for the count (hmc.hip: a streak of
    eight observations at the cap; the count comes back through host memory the device writes); GLMMR_MCML_HMC_SPEC=0 waits for
    every count.  The switch is read once per process: two child processes, the same draws bit for bit -- dense and sparse
    operator, step counts pinned at the cap (lambda large) and wandering below it."""
    import subprocess, sys, os
    code = """
import sys, numpy as np
sys.path.insert(0, %r)
from glmmrmcml_amd import api, synth
out = []
for gen, kw, lam, ms, ch in ((synth.geospatial, dict(n=300, seed=3), 5.0, 6, 40), (synth.geospatial, dict(n=300, seed=3), 0.3, 10, 40),
                             (synth.stepped_wedge, dict(ncl=7, nt=5, nind=40), 5.0, 6, 70), (synth.stepped_wedge, dict(ncl=7, nt=5, nind=40), 0.4, 10, 70)):
    d = gen(**kw)
    with api.Context(d["cov"], d["data"], d["eff_range"], d["Z"], d["X"], d["y"], d["family"], d["link"]) as ctx:
        ctx.update_L(d["theta"])
        diag, flags, probs = ctx.hmc_sample(d["beta"], 1.0, 40, ch * 2, lam, ms, 0.9, seed=5, chains=ch, want_trace=True)
        out += [ctx.get_u(), flags.astype(float), probs, np.array([diag["leapfrog_total"], diag["max_steps_used"]], float)]
np.savez(sys.argv[1], *out)
""" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))),)
    res = []
    for spec in ("1", "0"):
        f = str(tmp_path / ("spec%s.npz" % spec))
        env = dict(os.environ, GLMMR_MCML_HMC_SPEC=spec)
        subprocess.run([sys.executable, "-c", code, f], check=True, env=env, timeout=300)
        with np.load(f) as z:
            res.append([z[k] for k in sorted(z.files, key=lambda s: int(s.split("_")[1]))])
    assert len(res[0]) == len(res[1]) == 16
    for a, b in zip(res[0], res[1]):
        assert np.array_equal(a, b)


def test_dense_z_sampler_runs_the_direct_to_lds_kernel(orc):
    """a dense, non-identity Z (Householder reflector, bench.py --dense-z): ZL has no structural zeros, more than 16
    chains -> both products run dgemm_dlds_asm_kernel (asserted), chain by chain against the oracle"""
    from glmmrmcml_amd import api
    for n, Cn in ((200, 32), (330, 40)):
        d = synth.geospatial(n, seed=33)
        v = np.random.default_rng(n).standard_normal(n); v /= np.linalg.norm(v)
        d["Z"] = np.asfortranarray(np.eye(n) - 2.0 * np.outer(v, v))
        ctx, ZL, xb, fl, Lo = _setup(orc, d, api)
        warm, lam, ms, ta, seed, it = 12, 0.4, 6, 0.9, 99, 1
        diag, flags, probs = ctx.hmc_sample(d["beta"], d["sigma"], warm, Cn, lam, ms, ta, seed, chains=Cn, iter_idx=it,
                                            adapt=10, want_trace=True)
        assert ctx.last_kernels() == ("dlds", "dlds") and ctx.profile(enable=False)["operator"] == "dense"
        u = ctx.get_u()
        assert u.shape == (n, Cn)
        for c in range(Cn):
            so, fo, po, dg = orc.hmc_chain(xb, ZL, d["y"], d["sigma"], fl, warm, 1, lam, ms, ta, seed, chain_id=c,
                                           iter_idx=it, adapt=10)
            assert np.array_equal(flags[c], fo), (n, c)
            assert np.abs(probs[c] - po).max() < 1e-9
            uo = Lo @ so[:, 1:]
            assert np.abs(u[:, c:c + 1] - uo).max() < 1e-8 * max(1.0, np.abs(uo).max())
        ctx.close()
