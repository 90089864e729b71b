"""The N > 1 path of the PRODUCT (csrc/comm.hip, model.hip::model_mcnr_stats, drivers.hip: chain_offset =
rank * chains, every statistic all-reduced, the theta-step sharded over candidate thetas) on one GPU:

  * two contexts, rank 0 and rank 1 of world 2, driven from two host threads; the reduce hook sums the two
    contexts' device buffers (fixed order, barrier either side) -- what RCCL does between two GPUs.  mcml_full
    over 2 x C chains must reproduce the single-context run with 2C chains (src/mcml_full.cpp:83-145): beta,
    theta, sigma, each rank's u = its columns of the unsharded u and get_u_all = all of it (mcml_full.cpp:144-145);
    the number of collectives and their payloads are the ones DESIGN.md section 7 states (P*P+P+2 doubles per MCNR
    step, 2 per MCEM objective evaluation, per iteration one all-gather of the samples and one all-reduce of
    `world` candidate values per round of the theta-step);
  * one rank of an emulated 8-rank job (bench.py --as-rank-of 8): record and replay give the same fit, the replay
    evaluates an eighth of the candidates;
  * the native RCCL communicator (ncclAllReduce on the context's stream, no callback) on a 1-rank group: same
    numbers as no communicator, collectives counted.
"""
import ctypes as C
import threading

import numpy as np
import pytest

from glmmrmcml_amd import synth

pytestmark = pytest.mark.gpu


_HIP = None


def _hip():
    """the HIP runtime through ctypes: the hook moves its few doubles with plain (synchronous) hipMemcpy"""
    global _HIP
    if _HIP is None:
        _HIP = C.CDLL("libamdhip64.so")
        _HIP.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        _HIP.hipMemcpy.restype = C.c_int
    return _HIP


def _run_world2(d, kw, Cn):
    from glmmrmcml_amd import api
    args = (d["cov"], d["data"], d["eff_range"], d["Z"], d["X"], d["y"], d["family"], d["link"])
    bar = threading.Barrier(2, timeout=120)
    slots = [None, None]
    calls = [[], []]
    hip = _hip()

    def make_hook(rank):
        def hook(user, ptr, n):
            try:
                mine = np.zeros(n)
                assert hip.hipMemcpy(mine.ctypes.data, ptr, 8 * n, 2) == 0          # device -> host
                slots[rank] = mine
                bar.wait()
                tot = slots[0] + slots[1]            # the same order on both ranks: bit-identical sums
                assert hip.hipMemcpy(ptr, tot.ctypes.data, 8 * n, 1) == 0           # host -> device
                calls[rank].append(int(n))
                bar.wait()
                return 0
            except Exception as e:                   # never raise through the C frame
                print("hook failed:", repr(e), flush=True)
                bar.abort()
                return 1
        return hook

    out = [None, None]
    err = [None, None]

    def worker(rank):
        try:
            with api.Context(*args, rank=rank, world=2, reduce=make_hook(rank)) as ctx:
                r = ctx.mcml_full(d["start"], chains=Cn, m=Cn, **kw)
                r["u"] = ctx.get_u()
                r["comm"] = ctx.comm_stats()
                r["shard"] = ctx.shard_stats()
                r["ncalls"] = len(calls[rank])        # the exchanges of the fit itself (the exports below add theirs)
                r["u_all"] = ctx.get_u_all()          # the theta-step gathered them: no further collective
                assert ctx.comm_stats() == r["comm"] and ctx.shard_stats() == r["shard"]
                # the step exports on the sharded samples (column-sharded evaluations, every value all-reduced)
                vp = [r["sigma"]] if d["family"] == "gaussian" else []
                r["aic"] = ctx.aic_mcml(np.r_[r["beta"], vp], r["theta"])
                r["H"] = ctx.mcml_hess(np.r_[r["beta"], r["theta"], r["sigma"] if vp else 1.0], tol=1e-4)
                out[rank] = r
        except Exception as e:
            err[rank] = e
            bar.abort()

    th = [threading.Thread(target=worker, args=(r,)) for r in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join(600)
    assert err == [None, None], err
    return out, calls


@pytest.mark.parametrize("gen,gkw,mcnr", [(synth.cluster_rct, dict(ncl=8, nt=3, nind=6, seed=3), True),
                                          (synth.geospatial, dict(n=150, seed=9), True),
                                          (synth.cluster_rct, dict(ncl=8, nt=3, nind=6, seed=3), False)])
def test_world2_contexts_equal_one_context_with_all_chains(gen, gkw, mcnr):
    from glmmrmcml_amd import api
    d = gen(**gkw)
    Cn = 12
    # the optimisers run to convergence (default budget): a truncated trust-region run is path dependent, and the
    # sharded objective differs from the unsharded one in the last bits (sum over ranks of per-rank sums)
    kw = dict(mcnr=mcnr, maxiter=2, warmup=15, tol=1e-12, lambda_=0.3, maxsteps=6, target_accept=0.9, seed=4242)
    args = (d["cov"], d["data"], d["eff_range"], d["Z"], d["X"], d["y"], d["family"], d["link"])
    with api.Context(*args) as ctx:
        whole = ctx.mcml_full(d["start"], chains=2 * Cn, m=2 * Cn, **kw)
        whole["u"] = ctx.get_u()
        assert ctx.comm_stats()["calls"] == 0
    out, calls = _run_world2(d, kw, Cn)
    P = d["P"]
    for r in range(2):
        assert out[r]["iters"] == 2
        assert np.abs(out[r]["beta"] - whole["beta"]).max() < 2e-6 * max(1.0, np.abs(whole["beta"]).max())
        assert np.abs(out[r]["theta"] - whole["theta"]).max() < 2e-6
        assert abs(out[r]["sigma"] - whole["sigma"]) < 2e-6 * max(1.0, abs(whole["sigma"]))
        # rank r holds global chains [r*C, (r+1)*C): one draw per chain -> its columns of the unsharded u
        assert out[r]["u"].shape == (d["Q"], Cn)
        assert np.abs(out[r]["u"] - whole["u"][:, r * Cn:(r + 1) * Cn]).max() < 2e-5
        assert not out[r]["comm"]["native"]
        assert out[r]["comm"]["calls"] + out[r]["shard"]["gathers"] == out[r]["ncalls"]
        assert np.array_equal(out[r]["u_all"][:, r * Cn:(r + 1) * Cn], out[r]["u"])
        assert np.abs(out[r]["u_all"] - whole["u"]).max() < 2e-5
    assert np.array_equal(out[0]["beta"], out[1]["beta"]) and np.array_equal(out[0]["theta"], out[1]["theta"])
    assert np.array_equal(out[0]["u_all"], out[1]["u_all"])
    # aic_mcml / mcml_hess of the sharded job (src/mcml_optim.cpp:263-285,356-392) = the same exports of ONE context that
    # holds all the columns, at the same parameters
    assert out[0]["aic"] == out[1]["aic"] and np.array_equal(out[0]["H"], out[1]["H"])
    with api.Context(*args) as ctx:
        ctx.set_u(out[0]["u_all"])
        vp = [out[0]["sigma"]] if d["family"] == "gaussian" else []
        aic1 = ctx.aic_mcml(np.r_[out[0]["beta"], vp], out[0]["theta"])
        H1 = ctx.mcml_hess(np.r_[out[0]["beta"], out[0]["theta"], out[0]["sigma"] if vp else 1.0], tol=1e-4)
    assert abs(out[0]["aic"] - aic1) < 1e-9 * abs(aic1)
    assert np.abs(out[0]["H"] - H1).max() < 1e-4 * np.abs(H1).max()
    assert calls[0] == calls[1] and len(calls[0]) > 0
    calls = [c[:out[0]["ncalls"]] for c in calls]
    Q = d["Q"]
    ld = -(-Q // 32) * 32                               # leading dimension of the resident sample matrix
    gather = 2 * ld * Cn                                # the all-gather through a summing hook: both blocks
    sh = out[0]["shard"]
    width = 8 if gen is synth.geospatial else 2          # candidates per round: one per rank; a dense-block model: 8 / world per
    #                                                      rank, factorised side by side
    # per iteration: ONE all-gather of the samples (+ the 2-double agreement on the block width in front of it), then
    # one all-reduce of <= world candidate values per round of the theta-step; each rank evaluates half the candidates
    assert calls[0].count(gather) == 2 and sh["gathers"] == 2 and sh["gather_doubles"] == 2 * gather
    assert sh["theta_rounds"] >= 2 * 3 and sh["theta_evals_own"] + out[1]["shard"]["theta_evals_own"] == sh["theta_evals_all"]
    assert abs(sh["theta_evals_own"] - out[1]["shard"]["theta_evals_own"]) <= (width // 2) * sh["theta_rounds"]
    assert sh["theta_evals_all"] <= width * sh["theta_rounds"]
    small = [n for n in calls[0] if n != gather]
    if mcnr:
        # per iteration: the MCNR statistics (P*P + P + 2 doubles), the agreement on the block width (2), the rounds
        assert len(small) == 2 + 2 + sh["theta_rounds"] and small.count(P * P + P + 2) >= 2
        assert all(n <= max(width, P * P + P + 2) for n in small)
    else:
        assert len(small) > 2 + sh["theta_rounds"] and all(n <= max(width, 2) for n in small)   # + (sum, count) per MCEM evaluation


def test_emulated_rank_record_and_replay():
    """glmmr_mcml_dbg_emulate_world: one rank of an 8-rank job whose peers are copies of itself.  Record (all candidate
    thetas evaluated here) and replay (rank 0's share only, the rest from the record) must give the same fit, and the
    replay must evaluate rank 0's share, one candidate per round."""
    from glmmrmcml_amd import api
    d = synth.geospatial(n=150, seed=9)
    args = (d["cov"], d["data"], d["eff_range"], d["Z"], d["X"], d["y"], d["family"], d["link"])
    kw = dict(mcnr=True, maxiter=2, warmup=15, tol=1e-12, lambda_=0.3, maxsteps=6, target_accept=0.9, seed=4242,
              chains=8, m=8, maxfun=40)
    with api.Context(*args) as ctx:
        ctx.emulate_world(8, 1)
        rec = ctx.mcml_full(d["start"], **kw)
        s1 = ctx.shard_stats()
        urec = ctx.get_u_all()
        ctx.emulate_world(8, 2)
        rep = ctx.mcml_full(d["start"], **kw)
        s2 = ctx.shard_stats()
        urep = ctx.get_u_all()
        with pytest.raises(Exception, match="replay ran past its record"):
            ctx.mcml_optim(d["start"], mcnr=True, maxfun=8)
        ctx.emulate_world(1, 0)
    assert np.array_equal(rec["beta"], rep["beta"]) and np.array_equal(rec["theta"], rep["theta"])
    assert rec["sigma"] == rep["sigma"] and np.array_equal(urec, urep)
    assert urec.shape == (d["Q"], 64) and np.array_equal(urec[:, :8], urec[:, 56:])     # eight copies of the block
    rounds = s1["theta_rounds"]
    assert s1["theta_evals_own"] == rounds                               # rank 0 owns slot 0 of a round of 8
    assert s1["theta_evals_all"] <= 2 * 40 and s1["theta_evals_all"] > 4 * rounds       # rounds (of 8) are mostly full
    assert s2["theta_rounds"] - rounds == rounds and s2["theta_evals_own"] - s1["theta_evals_own"] == s1["theta_evals_own"]


def test_world_gt_1_without_exchange_is_an_error():
    from glmmrmcml_amd import api, _lib
    d = synth.cluster_rct(ncl=4, nt=2, nind=3, seed=1)
    args = (d["cov"], d["data"], d["eff_range"], d["Z"], d["X"], d["y"], d["family"], d["link"])
    with api.Context(*args, rank=0, world=2) as ctx:
        ctx.set_u(np.zeros((d["Q"], 3)))
        with pytest.raises(_lib.McmlError, match="neither an RCCL communicator"):
            ctx.mvn_ll(d["theta"])


def test_native_rccl_single_rank_group():
    """glmmr_mcml_ctx_comm_init_rccl on a 1-rank communicator: every statistic goes through ncclAllReduce on the
    context's stream (identity sum) -- same results as without it, collectives counted"""
    from glmmrmcml_amd import api
    d = synth.cluster_rct(ncl=8, nt=3, nind=6, seed=3)
    args = (d["cov"], d["data"], d["eff_range"], d["Z"], d["X"], d["y"], d["family"], d["link"])
    kw = dict(mcnr=True, maxiter=2, warmup=15, tol=1e-12, lambda_=0.3, maxsteps=6, target_accept=0.9, seed=7,
              maxfun=20, chains=10, m=10)
    with api.Context(*args) as ctx:
        base = ctx.mcml_full(d["start"], **kw)
        ub = ctx.get_u()
    with api.Context(*args) as ctx:
        ctx.comm_init_rccl(api.rccl_unique_id(), 0, 1)
        # the self-test dist.init_native_rccl runs before it commits to the native path (1 rank: the identity)
        assert np.array_equal(ctx.comm_allreduce([1.0, 1.0, 0.5]), [1.0, 1.0, 0.5])
        got = ctx.mcml_full(d["start"], **kw)
        ug = ctx.get_u()
        st = ctx.comm_stats()
        ll = ctx.mvn_ll(got["theta"])
        assert ctx.comm_stats()["calls"] == st["calls"] + 1
    assert st["native"] and st["calls"] >= 4 and st["doubles"] >= 2 * (d["P"] ** 2 + d["P"] + 2)
    assert np.array_equal(got["beta"], base["beta"]) and np.array_equal(got["theta"], base["theta"])
    assert np.array_equal(ug, ub) and np.isfinite(ll)
    # the batch theta-step on the same 1-rank communicator: the samples go through ncclAllGather, the candidate values
    # through ncclAllReduce; run to convergence it finds the sequential optimiser's theta
    kw2 = dict(kw); kw2.pop("maxfun")
    with api.Context(*args) as ctx:
        seq = ctx.mcml_full(d["start"], **kw2)
    with api.Context(*args) as ctx:
        ctx.comm_init_rccl(api.rccl_unique_id(), 0, 1)
        bat = ctx.mcml_full(d["start"], theta_batch=4, **kw2)
        sh = ctx.shard_stats()
        assert np.array_equal(ctx.get_u_all(), ctx.get_u())
    assert sh["gathers"] == 2 and sh["theta_rounds"] >= 6 and sh["theta_evals_all"] == sh["theta_evals_own"]
    assert np.abs(bat["theta"] - seq["theta"]).max() < 2e-6 and np.abs(bat["beta"] - seq["beta"]).max() < 2e-6
