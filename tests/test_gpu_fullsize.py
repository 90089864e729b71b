"""BASELINE-size checks through size-independent properties (the oracle would take hours at
these sizes), plus structures and edge cases the small parity cases do not reach.

Properties used:
  * L L' reconstructs D and L^-1 (L x) = x at Q = 2000 / 5000 (factorisation + TRSM);
  * mvn_ll is linear in the scale: D = t*R  =>  ll(t) = ll(1) - Q/2 log t - (1/t - 1) q/2, which
    ties evaluations at different theta together without a reference value;
  * the HMC sampler at cfg 2/3 sizes targets an exactly Gaussian posterior: the post-warm-up
    draws of all chains have the closed-form mean, and accept probabilities are in (0, 1];
  * sharded chains (two contexts with chain offsets) reproduce the unsharded chains exactly.
"""
import numpy as np
import pytest

from glmmrmcml_amd import synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("Q", [2000, 5000])
def test_cholesky_and_trsm_at_baseline_sizes(Q):
    from glmmrmcml_amd import api
    d = synth.geospatial(Q)
    rng = np.random.default_rng(1)
    with api.Context(d["cov"], d["data"], d["eff_range"]) as ctx:
        L = ctx.gen_D(d["theta"], chol=True)
        D = ctx.gen_D(d["theta"], chol=False)
        assert np.allclose(np.triu(L, 1), 0)
        x = rng.normal(size=Q)
        assert np.abs(L @ (L.T @ x) - D @ x).max() < 1e-11 * np.abs(D @ x).max()
        # u = L z  =>  quadratic form = |z|^2 ; logdet from diag(L)
        m = 7
        z = rng.normal(size=(Q, m))
        ctx.set_u(np.asfortranarray(L @ z))
        ll = ctx.mvn_ll(d["theta"])
        want = np.mean(-0.5 * Q * np.log(2 * np.pi) - np.log(np.diag(L)).sum() - 0.5 * (z ** 2).sum(0))
        assert abs(ll - want) < 1e-9 * abs(want)
        # scale linearity: theta0 -> t * theta0
        t = 1.7
        ll_t = ctx.mvn_ll([d["theta"][0] * t, d["theta"][1]])
        q = (z ** 2).sum(0).mean()
        assert abs(ll_t - (want - 0.5 * Q * np.log(t) - 0.5 * (1 / t - 1) * q)) < 1e-9 * abs(want)


@pytest.mark.parametrize("n,chains", [(2000, 256), (5000, 1024)])
def test_sampler_at_config_sizes_hits_the_gaussian_posterior(n, chains):
    from glmmrmcml_amd import api
    d = synth.geospatial(n)
    with api.Context(d["cov"], d["data"], d["eff_range"], d["Z"], d["X"], d["y"], d["family"], d["link"]) as ctx:
        ctx.update_L(d["theta"])
        diag, flags, probs = ctx.hmc_sample(d["beta"], d["sigma"], 60, chains, 5.0, 10, 0.9, seed=3,
                                            chains=chains, want_trace=True)
        u = ctx.get_u()
        L = ctx.gen_D(d["theta"], chol=True)
    assert u.shape == (n, chains)
    assert np.all(probs > 0) and np.all(probs <= 1) and 0.5 < diag["accept_rate"] <= 1
    # posterior mean of u = L v:  L S L'(y - xb)/s^2 with S = (I + L'L/s^2)^-1  (Z = I)
    s2 = d["sigma"] ** 2
    S = np.linalg.inv(np.eye(n) + L.T @ L / s2)
    mu_u = L @ (S @ (L.T @ (d["y"] - d["X"] @ d["beta"]))) / s2
    err = u.mean(1) - mu_u
    sd_u = np.sqrt(np.einsum("ij,jk,ik->i", L, S, L) / chains)
    # 60 warm-up proposals of 10 steps have not fully mixed: allow 6 sd + 10% of the signal
    assert np.mean(np.abs(err) < 6 * sd_u + 0.1 * np.abs(mu_u).max()) > 0.99


def test_sharded_chains_equal_unsharded_chains():
    from glmmrmcml_amd import api
    d = synth.geospatial(300, seed=4)
    args = (d["cov"], d["data"], d["eff_range"], d["Z"], d["X"], d["y"], d["family"], d["link"])
    with api.Context(*args) as ctx:
        ctx.update_L(d["theta"])
        ctx.hmc_sample(d["beta"], d["sigma"], 12, 16, 1.0, 6, 0.9, seed=8, chains=16)
        whole = ctx.get_u()
    parts = []
    for r in range(2):
        with api.Context(*args, rank=r, world=1) as ctx:
            ctx.update_L(d["theta"])
            ctx.hmc_sample(d["beta"], d["sigma"], 12, 8, 1.0, 6, 0.9, seed=8, chains=8, chain_offset=8 * r)
            parts.append(ctx.get_u())
    assert np.array_equal(np.concatenate(parts, axis=1), whole)


def test_mixed_block_structure(orc):
    """one cov matrix holding diagonal gr blocks, small ar1 blocks and a large dense fexp block at an
    odd offset: every mvn_ll code path in one call"""
    from glmmrmcml_amd import api
    rng = np.random.default_rng(5)
    rows, data = [], []
    for b in range(3):                         # 3 diagonal blocks of dim 1
        rows.append([b, 1, 1, 1, 0]); data.append([b + 1.0])
    nt = 5
    for b in range(4):                         # 4 gr*ar1 blocks of dim 5
        rows.append([3 + b, nt, 1, 1, 1]); rows.append([3 + b, nt, 3, 1, 2])
        data.append(np.r_[np.full(nt, b + 1.0), np.arange(1.0, nt + 1)])
    nd = 301                                   # dense fexp block, starts at the odd offset 23
    xy = rng.random((nd, 2))
    rows.append([7, nd, 7, 2, 3]); data.append(np.r_[xy[:, 0], xy[:, 1]])
    rows.append([8, 1, 1, 1, 0]); data.append([9.0])          # a trailing diagonal block
    cov = np.array(rows, dtype=np.int32)
    data = np.concatenate([np.atleast_1d(np.asarray(x, float)) for x in data])
    theta = np.array([0.3, 0.4, 0.6, 0.25, 0.1])
    Q = 3 + 4 * nt + nd + 1
    u = rng.normal(size=(Q, 11)) * 0.4
    got = api.mvn_ll(cov, data, np.zeros(len(rows)), theta, u)
    want = orc.mvn_ll(cov, data, np.zeros(len(rows)), theta, u)
    assert abs(got - want) < 1e-10 * abs(want)
    with api.Context(cov, data, np.zeros(len(rows))) as ctx:
        L = ctx.gen_D(theta, chol=True)
    Lo = orc.gen_D(cov, data, np.zeros(len(rows)), theta, chol=True)
    assert np.abs(L - Lo).max() < 1e-10


def test_edge_shapes(orc):
    from glmmrmcml_amd import api, _lib
    # Q = 1, one column
    cov = np.array([[0, 1, 1, 1, 0]], dtype=np.int32)
    assert abs(api.mvn_ll(cov, [1.0], [0.0], [0.5], [[0.3]]) - orc.mvn_ll(cov, [1.0], [0.0], [0.5], [[0.3]])) < 1e-14
    # n = 1 observation, P = 1
    d = synth.cluster_rct(ncl=2, nt=1, nind=1, seed=1)
    with api.Context(d["cov"], d["data"], d["eff_range"], d["Z"], d["X"][:, :2], d["y"], "binomial", "logit") as ctx:
        ctx.update_L(d["theta"])
        ctx.hmc_sample(d["beta"][:2], 1.0, 3, 2, 0.5, 3, 0.9, seed=1)
        assert ctx.get_u().shape == (d["Q"], 3)
    # zero columns of u is an error, not a crash
    with api.Context(cov, [1.0], [0.0]) as ctx:
        with pytest.raises(_lib.McmlError):
            ctx.mvn_ll([0.5])
    # dimensions that do not match
    with pytest.raises(_lib.McmlError):
        api.Context(d["cov"], d["data"], d["eff_range"], d["Z"][:, :-1], d["X"], d["y"], "binomial", "logit")


def test_longitudinal_poisson_reduced(orc):
    """config 5's structure (all-diagonal D, indicator Z, poisson-log) at reduced scale"""
    from glmmrmcml_amd import api
    from oracle import drivers
    d = synth.longitudinal(nsubj=60, nvisit=4, seed=6)
    args = (d["cov"], d["data"], d["eff_range"], d["Z"], d["X"], d["y"])
    rng = np.random.default_rng(2)
    L = orc.gen_D(d["cov"], d["data"], d["eff_range"], d["theta"], chol=True)
    u = np.asfortranarray(L @ rng.normal(size=(d["Q"], 33)))
    mod = drivers.Model(*args, d["family"], d["link"])
    got = api.mcml_optim(*args, u, d["family"], d["link"], d["start"], mcnr=True)
    want = drivers.mcml_optim(mod, u, d["start"], mcnr=True)
    assert np.abs(got["beta"] - want["beta"]).max() < 1e-8
    assert np.abs(got["theta"] - want["theta"]).max() < 2e-6
    H = api.mcml_hess(*args, u, d["family"], d["link"], np.r_[want["beta"], want["theta"], 1.0], tol=1e-4)
    Ho = drivers.mcml_hess(mod, u, np.r_[want["beta"], want["theta"], 1.0], tol=1e-4)
    assert np.abs(H - Ho).max() < 1e-4 * np.abs(Ho).max()


@pytest.mark.parametrize("chains", [1, 4, 11])
def test_streamed_and_mfma_products_agree_at_full_size(monkeypatch, chains):
    """n = Q = 5000, <= 16 chains: the streamed products (dgemm_skinny.h: 40 row blocks, 40 K chunks, the zero K ranges
    of the triangular ZL never launched into) against the 128-column MFMA tiles on the same chains -- identical
    accept/reject decisions, samples to rounding (chains = 1 is the reference's layout, Q x (m + 1))"""
    from glmmrmcml_amd import api
    d = synth.geospatial(5000)
    out = {}
    with api.Context(d["cov"], d["data"], d["eff_range"], d["Z"], d["X"], d["y"], d["family"], d["link"]) as ctx:
        ctx.update_L(d["theta"])
        for mode in ("skinny", "mfma"):
            if mode == "mfma":
                monkeypatch.setenv("GLMMR_MCML_SKINNY", "0")
            else:
                monkeypatch.delenv("GLMMR_MCML_SKINNY", raising=False)
            diag, flags, probs = ctx.hmc_sample(d["beta"], d["sigma"], 12, 4 * chains, 5.0, 10, 0.9, seed=9, chains=chains,
                                                want_trace=True)
            out[mode] = (ctx.get_u(), flags.copy(), probs.copy())
    monkeypatch.delenv("GLMMR_MCML_SKINNY", raising=False)
    assert out["skinny"][0].shape == (5000, 5 if chains == 1 else 4 * chains)
    assert np.array_equal(out["skinny"][1], out["mfma"][1])
    assert np.abs(out["skinny"][2] - out["mfma"][2]).max() < 1e-9
    assert np.abs(out["skinny"][0] - out["mfma"][0]).max() < 1e-8 * np.abs(out["mfma"][0]).max()


def test_graph_replay_of_the_factorisation_is_bit_identical_to_eager_launches():
    """mvn_ll at Q = 2000, m = 192, five evaluations per mode: eager launches (GLMMR_MCML_CHOL_GRAPH=0), the graph of
    the eager fork-join (=old) and the shipped graph captured from the two-chain schedule (default).  The first
    evaluation is always eager and the second is the capture; every tile receives its updates in panel order from
    kernels that accumulate k in order, so all fifteen values agree to the last bit."""
    import json, os, subprocess, sys
    code = ("import json, numpy as np\n"
            "from glmmrmcml_amd import api, synth\n"
            "d = synth.geospatial(2000, seed=5)\n"
            "ctx = api.Context(d['cov'], d['data'], d['eff_range'], d['Z'], d['X'], d['y'], d['family'], d['link'])\n"
            "ctx.set_u(np.asfortranarray(np.random.default_rng(1).standard_normal((2000, 192))))\n"
            "print(json.dumps([ctx.mvn_ll(d['theta'] * (1 + 0.02 * k)) for k in range(5)]))\n")
    out = {}
    for mode in ("0", "old", "2"):
        env = dict(os.environ, GLMMR_MCML_CHOL_GRAPH=mode)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600,
                           cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        assert r.returncode == 0, r.stderr[-2000:]
        out[mode] = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["0"] == out["old"] == out["2"]
    assert len(set(out["2"])) == 5 and all(np.isfinite(out["2"]))


@pytest.mark.parametrize("n,m", [(2000, 256), (5000, 1024)])
def test_whole_mcml_iterations_theta_step_on_the_graph_equals_eager_evaluation(n, m, tmp_path):
    """BASELINE configs 2 and 3 at full size, two whole mcml_full iterations (sampler -> MCNR -> theta-step of at most 40
    evaluations, eight candidates per round factorised side by side on the replayed graph with the m sample columns
    appended -> L refresh; src/mcml_full.cpp:83-140): every objective value the last theta-step saw is re-evaluated on
    the same samples ONE AT A TIME by EAGER launches (GLMMR_MCML_CHOL_GRAPH=0, a fresh process) and must agree to
    rounding -- 1e-12 relative: the batch regroups its trailing updates into K = 1024 passes, the single evaluation does
    not (mcmldmatrix.h:23-41); the fit stays in a band round the generating values (theta = (0.25, 0.1), sigma = 1,
    beta = 1)."""
    import json, os, subprocess, sys
    from glmmrmcml_amd import api
    d = synth.geospatial(n, seed=20240601)
    kw = dict(mcnr=True, m=m, warmup=100, tol=0.0, lambda_=5.0, maxsteps=10, target_accept=0.9, seed=20240601, chains=m,
              maxfun=40)
    with api.Context(d["cov"], d["data"], d["eff_range"], d["Z"], d["X"], d["y"], d["family"], d["link"]) as ctx:
        ctx.theta_log(enable=True)
        ctx.mcml_full(d["start"], maxiter=1, **kw)
        n1 = ctx.theta_log(enable=True).shape[0]         # evaluations of the first theta-step (the same in the run below)
        r = ctx.mcml_full(d["start"], maxiter=2, **kw)
        log = ctx.theta_log(enable=False)
        u = ctx.get_u()
        sh = ctx.shard_stats()
    assert r["iters"] == 2 and u.shape == (n, m) and np.all(np.isfinite(u))
    assert 8 <= n1 <= 40 and n1 < log.shape[0] <= 80 and log.shape[1] == 3     # at most 40 evaluations per theta-step
    # (a candidate at which D is not numerically positive definite is counted but has no value to log)
    assert n1 + log.shape[0] <= sh["theta_evals_all"] <= n1 + log.shape[0] + 12 and sh["theta_rounds"] <= 3 * 7
    last = log[n1:]
    assert 0.1 < r["theta"][0] < 0.6 and 0.03 < r["theta"][1] < 0.3 and 0.7 < r["sigma"] < 1.3 and 0.0 < r["beta"][0] < 2.0
    assert np.all(np.isfinite(last))
    np.save(tmp_path / "u.npy", u); np.save(tmp_path / "th.npy", last[:, :2])
    code = ("import json, numpy as np\n"
            "from glmmrmcml_amd import api, synth\n"
            "d = synth.geospatial(%d, seed=20240601)\n"
            "ctx = api.Context(d['cov'], d['data'], d['eff_range'])\n"
            "ctx.set_u(np.load(r'%s'))\n"
            "print(json.dumps([ctx.mvn_ll(t) for t in np.load(r'%s')]))\n" % (n, tmp_path / "u.npy", tmp_path / "th.npy"))
    env = dict(os.environ, GLMMR_MCML_CHOL_GRAPH="0")
    rr = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900,
                        cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert rr.returncode == 0, rr.stderr[-2000:]
    eager = np.array(json.loads(rr.stdout.strip().splitlines()[-1]))
    assert np.allclose(eager, last[:, 2], rtol=1e-12, atol=0), np.abs(eager - last[:, 2]).max()
