"""Chain sharding over the GPUs of one node.

One process per GPU.  Independent HMC chains / sample columns shard naturally:
rank r owns global chains [r*C, (r+1)*C); ZL, X, y and L are replicated; the RNG
streams are keyed by the GLOBAL chain id, so a chain's draws do not depend on
how many ranks there are.  Exchanges: an all-reduce (sum, f64) of the per-chain
sufficient statistics -- P*P + P + 2 doubles for the MCNR beta-step, 2 doubles
(sum, count) per objective evaluation of the MCEM step.  The theta-step shards
over CANDIDATES instead (csrc/drivers.hip d_optim_sharded): one all-gather of the
sample columns per iteration, then every rank evaluates one candidate theta per
round on all columns and one all-reduce per round carries the values.  The small
payloads are latency-bound: ONE collective per exchange, on the stream the kernels
run on.  torch.distributed's "nccl" backend is RCCL on ROCm and runs over xGMI
between the GPUs of a node.
"""
import ctypes as C

import numpy as np


def shard(total, world, rank):
    """[lo, hi) of `total` units for `rank` of `world`: contiguous, sizes differ by <= 1"""
    base, rem = divmod(int(total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def chain_offset(chains_per_rank, rank):
    return int(chains_per_rank) * int(rank)


def combine_mcnr(stats_list, P):
    """what the all-reduce of the MCNR statistics computes, on host arrays:
    stats = [sum X'W_iX (P*P) | sum X'W_i d_i r_i (P) | sum sigma_i | count]"""
    tot = np.sum(np.asarray(stats_list, dtype=float), axis=0)
    m = tot[P * P + P + 1]
    XtWX = tot[:P * P].reshape(P, P, order="F") / m
    XtWr = tot[P * P:P * P + P] / m
    return np.linalg.solve(XtWX, XtWr), tot[P * P + P] / m


class _DevArray:
    """a raw device pointer as a __cuda_array_interface__ producer (no copy)"""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<f8", "data": (int(ptr), False),
                                         "version": 2}


def make_reduce_hook(group=None):
    """reduce hook for glmmr_mcml_dev_opts: sums `n` doubles at a device pointer over the
    process group, in place, with torch.distributed (backend nccl = RCCL over xGMI)."""
    import torch
    import torch.distributed as dist

    def hook(user, ptr, n):
        try:
            t = torch.as_tensor(_DevArray(ptr, n), device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
            torch.cuda.current_stream().synchronize()
            return 0
        except Exception as e:        # never let an exception cross the C boundary
            print("glmmrmcml_amd.dist: reduce hook failed:", e, flush=True)
            return 1
    return hook


def init_native_rccl(ctx, rank, world, group=None, device=None):
    """give `ctx` its own RCCL communicator (csrc/comm.hip): rank 0 makes the id, torch.distributed only carries
    the 128 bytes to the other ranks; afterwards the library all-reduces on its own stream with no callback.
    Every step that could fail on one rank only is agreed on first (ncclCommInitRank is itself a collective: a
    rank that skipped it would leave the others waiting).  Raises on EVERY rank or on none."""
    import torch
    import torch.distributed as dist
    from . import api
    dev = device if device is not None else ("cuda" if dist.get_backend(group) == "nccl" else "cpu")

    def agreed(ok, what, err):
        """every rank reaches this all-reduce whatever happened locally; raises on all ranks or on none"""
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        if int(flag.item()) != 1:
            raise RuntimeError(what + ("" if ok else ": %r" % (err,)))

    # 1. everything a rank can check on its own: librccl resolves (rank 0's id is the one that is used), rank / world
    ok, uid, err = True, bytes(128), None
    try:
        uid = api.rccl_unique_id()
        if not (0 <= int(rank) < int(world)):
            raise ValueError("rank %r of %r" % (rank, world))
    except Exception as e:                  # noqa: BLE001
        ok, err = False, e
    agreed(ok, "librccl is not usable on every rank", err)
    t = torch.tensor(list(uid), dtype=torch.uint8, device=dev)
    dist.broadcast(t, src=0, group=group)
    # 2. ncclCommInitRank: itself a collective -- after step 1 only the collective can fail, and the outcome is shared
    ok, err = True, None
    try:
        ctx.comm_init_rccl(bytes(t.cpu().tolist()), rank, world)
    except Exception as e:                  # noqa: BLE001
        ok, err = False, e
    agreed(ok, "ncclCommInitRank failed on some rank", err)
    # 3. self-test of the new communicator on known values before anything depends on it
    ok, err = True, None
    try:
        got = ctx.comm_allreduce([rank + 1.0, 1.0, 0.5 * (rank + 1.0)])
        tri = world * (world + 1) / 2.0
        if not (got[0] == tri and got[1] == float(world) and got[2] == 0.5 * tri):
            ok, err = False, RuntimeError("native all-reduce returned %r on rank %d of %d" % (got.tolist(), rank, world))
    except Exception as e:                  # noqa: BLE001
        ok, err = False, e
    agreed(ok, "native RCCL all-reduce failed its self-test", err)
