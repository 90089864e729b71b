"""The N > 1 path on CPU: world_size 2, gloo.

What shards: chains / sample columns.  What is exchanged: ONE all-reduce (sum, f64) of the
per-chain sufficient statistics.  The GPU kernels cannot run here, so each rank's local
statistics come from the CPU oracle; what is under test is the sharding plan
(glmmrmcml_amd.dist), the reduction algebra the product uses (stats | count, then finalise),
and that the RNG streams are keyed by the GLOBAL chain id so results do not depend on the
number of ranks."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as tdist
import torch.multiprocessing as mp

from glmmrmcml_amd import dist as gdist
from glmmrmcml_amd import synth


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    tdist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import drivers, oracle as orc
    d = synth.cluster_rct(ncl=6, nt=3, nind=5, seed=4)
    fl = orc.flink(d["family"], d["link"])
    P, Q = d["P"], d["Q"]
    C_total, warm, lam, ms, seed, it = 8, 10, 0.3, 6, 99, 1
    lo, hi = gdist.shard(C_total, world, rank)
    assert gdist.chain_offset(C_total // world, rank) == lo
    mod = drivers.Model(d["cov"], d["data"], d["eff_range"], d["Z"], d["X"], d["y"], d["family"], d["link"])
    # this rank's chains, global ids lo..hi-1
    u_loc, _ = drivers.sample(mod, d["beta"], d["theta"], 1.0, warm, hi - lo, lam, ms, 0.9, seed, it,
                              chains=hi - lo, chain_offset=lo)
    # MCNR statistics of the local columns: [X'WX | X'Wr | sum sigma | count]
    r = orc.mcnr(d["X"], d["Z"], d["y"], u_loc, d["beta"], 1.0, d["family"], d["link"])
    stats = np.r_[r["XtWX"].ravel(order="F"), r["XtWr"], r["sigma_sum"], float(u_loc.shape[1])]
    t = torch.from_numpy(stats.copy())
    tdist.all_reduce(t, op=tdist.ReduceOp.SUM)
    inc, sigma = gdist.combine_mcnr([t.numpy()], P)
    # theta-step objective: (sum over local columns, count) all-reduced
    ll_loc = orc.mvn_ll(d["cov"], d["data"], d["eff_range"], d["theta"], u_loc) * u_loc.shape[1]
    t2 = torch.tensor([ll_loc, float(u_loc.shape[1])], dtype=torch.float64)
    tdist.all_reduce(t2)
    # gather u for the final comparison (mcml_full returns u, mcml_full.cpp:145)
    parts = [torch.zeros(Q, C_total // world, dtype=torch.float64) for _ in range(world)]
    tdist.all_gather(parts, torch.from_numpy(np.ascontiguousarray(u_loc)))
    if rank == 0:
        out["beta"] = d["beta"] + inc; out["sigma"] = sigma
        out["mvn"] = float(t2[0] / t2[1]); out["u"] = np.concatenate([p.numpy() for p in parts], axis=1)
    tdist.destroy_process_group()


def test_two_ranks_equal_one_rank(orc):
    from oracle import drivers
    world = 2
    mgr = mp.Manager(); out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    d = synth.cluster_rct(ncl=6, nt=3, nind=5, seed=4)
    mod = drivers.Model(d["cov"], d["data"], d["eff_range"], d["Z"], d["X"], d["y"], d["family"], d["link"])
    u, _ = drivers.sample(mod, d["beta"], d["theta"], 1.0, 10, 8, 0.3, 6, 0.9, 99, 1, chains=8)
    assert np.array_equal(out["u"], u)                 # chains do not depend on the rank count
    r = orc.mcnr(d["X"], d["Z"], d["y"], u, d["beta"], 1.0, d["family"], d["link"])
    assert np.allclose(out["beta"], r["beta"], rtol=1e-12, atol=1e-14)
    assert abs(out["sigma"] - r["sigma"]) < 1e-13
    assert abs(out["mvn"] - orc.mvn_ll(d["cov"], d["data"], d["eff_range"], d["theta"], u)) < 1e-10


def test_shard_plan():
    for total, world in ((1024, 8), (10, 3), (7, 8), (1, 1)):
        spans = [gdist.shard(total, world, r) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == total
        assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1


def test_reduce_hook_wraps_a_device_pointer_without_copy():
    """the hook hands torch a __cuda_array_interface__ view of the library's buffer"""
    arr = gdist._DevArray(0x7f0000001000, 5)
    cai = arr.__cuda_array_interface__
    assert cai["shape"] == (5,) and cai["typestr"] == "<f8" and cai["data"] == (0x7f0000001000, False)


# ---- the candidate-sharded theta-step (csrc/drivers.hip d_optim_sharded) over gloo ------------------------------------
# The product's optimiser (bobyqa_batch, host code of the library) runs on BOTH ranks; a round's candidates are split
# j mod world; every rank evaluates its share on ALL sample columns -- here with the CPU oracle's mvn_ll, after an
# all_gather of the ranks' columns -- and one all-reduce of a vector that is zero outside the owner's slots carries the
# values.  Under test: both ranks propose identical points in every round without any further agreement, the exchange
# pattern (one gather, one small all-reduce per round), and that the result is the single-process optimum.
def _theta_worker(rank, world, port, out):
    import ctypes as C
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    tdist.init_process_group("gloo", rank=rank, world_size=world)
    from glmmrmcml_amd import _lib
    from oracle import oracle as orc
    d = synth.geospatial(40, seed=3)
    rng = np.random.default_rng(100)                      # all columns, then this rank's share (what its sampler left)
    D = orc.gen_D(d["cov"], d["data"], d["eff_range"], d["theta"])
    U_all = np.linalg.cholesky(D) @ rng.standard_normal((40, 12))
    lo, hi = gdist.shard(12, world, rank)
    parts = [torch.zeros(40, 12 // world, dtype=torch.float64) for _ in range(world)]
    tdist.all_gather(parts, torch.from_numpy(np.ascontiguousarray(U_all[:, lo:hi])))          # ONE gather per theta-step
    U = np.concatenate([p.numpy() for p in parts], axis=1)
    assert np.array_equal(U, U_all)
    rounds = []

    def fb(Xp, n, k, Fp, user):
        X = np.array([[Xp[j * n + i] for i in range(n)] for j in range(k)])
        vals = np.zeros(k)
        for j in range(k):
            if j % world == rank:
                try:
                    vals[j] = -orc.mvn_ll(d["cov"], d["data"], d["eff_range"], np.exp(X[j]), U)
                except RuntimeError:
                    vals[j] = np.inf
        t = torch.from_numpy(vals)
        tdist.all_reduce(t, op=tdist.ReduceOp.SUM)          # every slot is zero on all ranks but its owner
        rounds.append((X.copy(), t.numpy().copy()))
        for j in range(k):
            Fp[j] = float(t[j])
        return 0
    dp = C.POINTER(C.c_double)
    cb = C.CFUNCTYPE(C.c_int, dp, C.c_int, C.c_int, dp, C.c_void_p)(fb)
    z0 = np.log(np.asarray(d["theta"], float) * np.array([1.3, 0.8])); lo_b = np.full(2, np.log(1e-6)); up_b = np.full(2, np.inf)
    x = np.zeros(2); f = C.c_double(); nf = C.c_int(); rd = C.c_int()
    _lib.check(_lib.lib().glmmr_mcml_dbg_bobyqa_rounds(cb, None, 2, z0.ctypes.data_as(dp), lo_b.ctypes.data_as(dp),
                                                       up_b.ctypes.data_as(dp), C.c_double(0.25), C.c_double(1e-7), 0, world,
                                                       x.ctypes.data_as(dp), C.byref(f), C.byref(nf), C.byref(rd)))
    out[rank] = dict(theta=np.exp(x), f=f.value, nf=nf.value, rounds=rd.value,
                     pts=[r[0] for r in rounds], vals=[r[1] for r in rounds])
    tdist.destroy_process_group()


def test_candidate_sharded_theta_step_over_gloo(orc):
    import ctypes as C
    from glmmrmcml_amd import _lib
    world = 2
    mgr = mp.Manager(); out = mgr.dict()
    mp.spawn(_theta_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    a, b = out[0], out[1]
    # both ranks ran the same rounds on the same values and ended at the same point, with nothing exchanged but the values
    assert a["rounds"] == b["rounds"] == len(a["pts"]) and a["nf"] == b["nf"]
    assert all(np.array_equal(p, q) for p, q in zip(a["pts"], b["pts"])) and all(np.array_equal(p, q) for p, q in zip(a["vals"], b["vals"]))
    assert np.array_equal(a["theta"], b["theta"]) and a["f"] == b["f"]
    assert max(p.shape[0] for p in a["pts"]) <= world and a["rounds"] < a["nf"]
    # ... which is the optimum the sequential optimiser finds on all columns in one process
    d = synth.geospatial(40, seed=3)
    D = orc.gen_D(d["cov"], d["data"], d["eff_range"], d["theta"])
    U = np.linalg.cholesky(D) @ np.random.default_rng(100).standard_normal((40, 12))
    dp = C.POINTER(C.c_double)

    def obj(xp, n, user):
        try:
            return -orc.mvn_ll(d["cov"], d["data"], d["eff_range"], np.array([xp[0], xp[1]]), U)
        except RuntimeError:
            return 1e300
    cb = C.CFUNCTYPE(C.c_double, dp, C.c_int, C.c_void_p)(obj)
    x0 = np.asarray(d["theta"], float) * np.array([1.3, 0.8]); lo = np.full(2, 1e-6); up = np.full(2, np.inf)
    x = np.zeros(2); f = C.c_double(); nf = C.c_int()
    _lib.check(_lib.lib().glmmr_mcml_dbg_bobyqa(cb, None, 2, x0.ctypes.data_as(dp), lo.ctypes.data_as(dp), up.ctypes.data_as(dp),
                                                C.c_double(0.0), C.c_double(1e-9), 0, x.ctypes.data_as(dp), C.byref(f), C.byref(nf)))
    assert abs(a["f"] - f.value) < 1e-9 * abs(f.value) and np.abs(a["theta"] - x).max() < 2e-6 * np.abs(x).max()
