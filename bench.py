#!/usr/bin/env python
"""bench.py -- BASELINE.json's metric on BASELINE.json's config.

metric : MCML iters/sec x m chains (simlik evals/sec)
step   : one MCML iteration = one pass of the `while` body of mcml_full
         (src/mcml_full.cpp:83-140): draw the samples (HMC) + beta-step (MCNR) +
         theta-step (BOBYQA on the MVN log-likelihood) + L refresh.
workload (every N): BASELINE config 3 as written -- Gaussian geospatial, n = Q = 5000,
         ~(1|fexp(x,y)), dense Sigma, MCNR, m = 1024 chains IN TOTAL, sharded over the N
         GPUs (STRONG scaling: rank r owns global chains [r m/N, (r+1) m/N); ZL/X/y/L
         replicated; one RCCL all-reduce of the statistics per evaluation), HMC warmup
         100 + 1 draw per chain, 10 leapfrog steps, theta-step budget 40 objective
         evaluations.  Synthetic data, seed 20240601.
         --weak keeps m = 1024 chains PER GPU instead ("scaling": "weak").
         --chains C overrides the chains per GPU (e.g. --gpus 1 --chains 128 = what one
         rank of the 8-GPU job runs).
         --dense-z replaces Z = I by a dense orthogonal-like Z (ZL dense: no zero
         skipping) -- a second, clearly labelled workload, not the metric's.
         --as-rank-of N (with --gpus 1): what ONE rank of the N-GPU job executes, on one GPU:
         1024 / N chains, the theta-step sharded over candidate thetas with this rank's share
         of every round, the peers emulated as copies of this rank (their candidate values
         come from an untimed recording run of the same iterations; include/glmmr_mcml_c.h
         glmmr_mcml_dbg_emulate_world).  The time of the collectives themselves is NOT in it.
Inputs are resident in HBM before the timed region starts.

Launch: python bench.py --gpus 1 --steps K --warmup W      (single GPU)
        python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6      # MI355X FP64 matrix, vendor figure (SURVEY.md 8d / BASELINE.md)

CFG = dict(n=5000, chains_total=1024, chains_per_gpu=1024, hmc_warmup=100, max_steps=10, lambda_=5.0, target_accept=0.9,
           theta_maxfun=40, seed=20240601)


def cpu_baseline(d, cfg, budget_props=30):
    """The CPU path timed on this box's host cores on a BOUNDED sample, extrapolated to one MCML
    iteration with the reference's own operation counts.  Checker code only (oracle/ + numpy)."""
    import scipy.linalg as sla
    from oracle import oracle as orc
    try:
        from threadpoolctl import threadpool_limits
    except Exception:                                   # pragma: no cover
        threadpool_limits = None
    threads = orc.num_threads()
    n = cfg["n"]; m = cfg["chains_total_run"]; W = cfg["hmc_warmup"]; E = cfg["theta_maxfun"]
    ctxmgr = threadpool_limits(limits=threads) if threadpool_limits else None
    if ctxmgr is not None:
        ctxmgr.__enter__()
    try:
        t0 = time.time(); D = orc.gen_D(d["cov"], d["data"], d["eff_range"], d["theta"]); t_build = time.time() - t0
        t0 = time.time(); L = np.linalg.cholesky(D); t_chol = time.time() - t0          # Eigen LLT's job
        rng = np.random.default_rng(1)
        ucol = L @ rng.standard_normal(n)
        t0 = time.time(); sla.solve_triangular(L, ucol, lower=True); t_fsub = time.time() - t0
        U = np.asfortranarray(L @ rng.standard_normal((n, 64)))
        t0 = time.time(); sla.solve_triangular(L, U, lower=True); t_trsm64 = time.time() - t0
        t0 = time.time(); ZU = d["Z"] @ U; t_zu64 = time.time() - t0                    # Z*u, 64 columns
        ZL = np.asfortranarray(d["Z"] @ L) if cfg.get("dense_z") else np.asfortranarray(L)  # Z = I
        xb = d["X"] @ d["beta"]
        t0 = time.time()
        orc.hmc_chain(xb, ZL, d["y"], d["sigma"], 7, budget_props - 1, 1, cfg["lambda_"], cfg["max_steps"],
                      cfg["target_accept"], cfg["seed"])
        t_prop = (time.time() - t0) / budget_props
    finally:
        if ctxmgr is not None:
            ctxmgr.__exit__(None, None, None)
    t_zu = t_zu64 * m / 64.0
    t_refresh = t_build + t_chol + t_zu64 * n / 64.0                                     # genD + Z*L
    # reference-faithful (what mcml_full does): one sequential chain of W + m proposals
    # (mhmcmc.h:121-157); MCNR re-multiplies Z*u once per sample (mcmlmodel.h:121, defect D3);
    # every theta evaluation rebuilds + refactorises the block once per column (mcmldmatrix.h:59, D2)
    t_faithful = (W + m) * t_prop + m * t_zu + E * (m + 1) * (t_build + t_chol + t_fsub) + t_refresh
    # algorithmically fair: factor once per theta, Z*u cached
    t_fair = (W + m) * t_prop + t_zu + E * (t_build + t_chol + t_trsm64 * (m + 1) / 64.0) + t_refresh
    return {
        "value": m / t_faithful, "unit": "simlik evals/s", "cores": int(threads), "kind": "port",
        "value_fair": m / t_fair,
        "sample": ("oracle C restatement (OpenMP) + numpy/LAPACK standing in for Eigen, %d threads; timed: %d HMC "
                   "proposals at n=Q=%d (%.3f s each), 1 covariance build (%.2f s), 1 Cholesky (%.2f s), 1 forward-sub "
                   "(%.3f s), TRSM and Z*u on 64 columns; extrapolated to one MCML iteration = %d sequential proposals "
                   "+ MCNR + %d theta evaluations + L refresh. value = reference-faithful (per-column refactorisation, "
                   "Z*u per sample: %.3g s/iter); value_fair = factor once per theta, cached Z*u (%.3g s/iter)"
                   % (threads, budget_props, n, t_prop, t_build, t_chol, t_fsub, W + m, E, t_faithful, t_fair)),
    }


HBM_PEAK_TBS = 8.0                # MI355X HBM3E (MI355X_MICROARCH.md)


def other_configs(api, synth, stream, iters=3):
    """BASELINE configs 2, 4 and 5 on this GPU, `iters` whole mcml_full iterations each after one untimed iteration
    (HMC warm-up 100 + 1 draw per chain, <= 10 leapfrog steps, optimiser budget 40): ms per iteration, simlik evals/s and
    the roofline fraction of the configuration's dominant kernel pair (the two HMC products), timed with HIP events on
    the library's stream like the headline line.  Config 5 also times mcml_hess (its 'Hessian SE')."""
    import torch
    out = {}
    specs = [("cfg2", "gaussian geospatial n=Q=2000 fexp, MCEM, m=256", lambda: synth.geospatial(2000, seed=1), 256, False, 5.0),
             ("cfg4", "binomial stepped-wedge 40 cl x 8 t x 50 ind (n=16000, Q=320, gr*ar1 blocks of 8), MCNR, m=512",
              lambda: synth.stepped_wedge(40, 8, 50), 512, True, 0.5),
             ("cfg5", "poisson longitudinal 2000 subjects x 10 visits (n=20000, Q=22000, diagonal D), MCNR, m=1024",
              lambda: synth.longitudinal(2000, 10), 1024, True, 0.5)]
    for key, desc, gen, m, mcnr, lam in specs:
        try:
            d = gen()
            with api.Context(d["cov"], d["data"], d["eff_range"], d["Z"], d["X"], d["y"], d["family"], d["link"],
                             stream=stream) as ctx:
                kw = dict(mcnr=mcnr, m=m, warmup=100, tol=0.0, verbose=False, lambda_=lam, maxsteps=10, target_accept=0.9,
                          seed=7, chains=m, maxfun=40)
                ctx.profile(enable=True, reset=True)             # markers on for the untimed iteration too: their first use
                ctx.mcml_full(d["start"], maxiter=1, **kw)        # (and whatever else is first-time) stays out of the timing
                if os.environ.get("GLMMR_BENCH_SETTLE"):           # experiment: idle time before the timed repetitions
                    torch.cuda.synchronize(); time.sleep(float(os.environ["GLMMR_BENCH_SETTLE"]))
                # two repetitions of the same `iters` iterations, both reported: the small configurations are bound by
                # launches and host wake-ups, and now and then a whole repetition runs ~40 % slower with the same kernel
                # times ("run-to-run jitter"); ms_per_iter is the faster one
                reps = []
                ph = []
                for _ in range(2):
                    ctx.profile(enable=True, reset=True)
                    api.phase_ms(enable=True, reset=True)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    r_ = ctx.mcml_full(d["start"], maxiter=iters, **kw)
                    torch.cuda.synchronize()
                    dt_ = time.perf_counter() - t0
                    p_ = ctx.profile(enable=False)
                    ph.append({k: round(v / iters, 2) for k, v in api.phase_ms(enable=False).items()})
                    reps.append(dt_)
                    if dt_ <= min(reps):
                        dt, r, p = dt_, r_, p_
                nl = p["fwd_n"] + p["bwd_n"]
                ks = (p["fwd_ms"] + p["bwd_ms"]) * 1e-3
                n, Q = d["n"], d["Q"]
                if p["operator"] == "sparse":
                    # algorithmic bytes per product: forward 8 (nC [S] + QC [X]), backward 8 (nC + 4 QC)
                    bytes_alg = (p["fwd_n"] * 8.0 * (n * m + Q * m) + p["bwd_n"] * 8.0 * (n * m + 4 * Q * m))
                    roof = {"bound": "hbm", "kernel": "k_cm_forward / k_cm_backward (sparse ZL operator, chain-major)",
                            "achieved": bytes_alg / ks / 1e12 if nl else 0.0, "peak": HBM_PEAK_TBS, "unit": "TB/s"}
                else:
                    ex = p["fwd_flops"] * p["fwd_n"] + p["bwd_flops"] * p["bwd_n"]
                    roof = {"bound": "mfma", "kernel": "dgemm_band_kernel + k_band_reduce (HMC products, FP64 MFMA)",
                            "achieved": ex / ks / 1e12 if nl else 0.0, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s"}
                roof["frac"] = roof["achieved"] / roof["peak"]
                roof["avg_launch_us"] = ks / max(1, nl) * 1e6
                roof["share_of_iteration"] = ks / max(1, nl) * (p["fwd_n_all"] + p["bwd_n_all"]) / dt
                rec = {"workload": desc, "ms_per_iter": dt / iters * 1e3, "ms_per_iter_reps": [x / iters * 1e3 for x in reps], "phases_ms_per_iter_reps": ph,
                       "evals_per_s": m * iters / dt, "iters": iters,
                       "roofline": roof, "beta": [float(x) for x in r["beta"]], "theta": [float(x) for x in r["theta"]],
                       "sigma": float(r["sigma"]), "accept_rate": r["accept_rate"]}
                rec["fit_ok"] = bool(np.all(np.isfinite(r["beta"])) and np.all(np.isfinite(r["theta"])) and
                                     np.all(np.asarray(r["theta"]) > 0) and
                                     np.all(np.abs(np.asarray(r["theta"]) - d["theta"]) < 0.6 * np.maximum(d["theta"], 0.15)))
                if key == "cfg5":
                    start = np.r_[r["beta"], r["theta"], 1.0]
                    torch.cuda.synchronize(); t0 = time.perf_counter()
                    H = ctx.mcml_hess(start, tol=1e-4)
                    torch.cuda.synchronize()
                    rec["mcml_hess_ms"] = (time.perf_counter() - t0) * 1e3
                    rec["fit_ok"] = bool(rec["fit_ok"] and np.all(np.isfinite(H)))
                out[key] = rec
        except Exception as e:                       # never blocks the headline number
            out[key] = {"workload": desc, "error": repr(e)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--n", "--nobs", dest="n", type=int, default=CFG["n"], help="override n = Q (debug only; invalidates the metric)")
    ap.add_argument("--chains", type=int, default=0, help="chains per GPU (default: 1024 / gpus; with --weak 1024)")
    ap.add_argument("--weak", action="store_true", help="weak scaling: 1024 chains per GPU")
    ap.add_argument("--reduce", choices=("native", "torch"), default="native",
                    help="N > 1: the library's own RCCL communicator (default) or the torch.distributed hook")
    ap.add_argument("--dense-z", action="store_true", help="dense (non-identity) Z: ZL dense, no zero skipping")
    ap.add_argument("--as-rank-of", type=int, default=0, help="time one rank of an N-rank job on one GPU (peers emulated)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip BASELINE configs 2, 4, 5 after the timed region")
    args = ap.parse_args()

    import torch
    from glmmrmcml_amd import api, dist as gdist, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d"
                             % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product has no CPU fallback")
    # GLMMR_MCML_DIST_BACKEND=gloo: rehearsal of the N > 1 path on a box with fewer GPUs than ranks (ranks
    # share devices, gloo carries the all-reduce of the CUDA tensors); the real run is nccl = RCCL over xGMI
    backend = os.environ.get("GLMMR_MCML_DIST_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    hook = None
    if world > 1:
        import torch.distributed as tdist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            tdist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            tdist.init_process_group(backend=backend)
        if args.reduce == "torch" or backend != "nccl":
            hook = gdist.make_reduce_hook()

    cfg = dict(CFG); cfg["n"] = args.n
    emu = args.as_rank_of if args.as_rank_of > 1 else 0
    if emu:
        assert world == 1, "--as-rank-of emulates the peers on ONE GPU: launch it with --gpus 1"
        assert CFG["chains_total"] % emu == 0
        if args.chains <= 0:
            args.chains = CFG["chains_total"] // emu
    if args.chains > 0:
        cfg["chains_per_gpu"] = args.chains
    elif args.weak:
        cfg["chains_per_gpu"] = CFG["chains_total"]
    else:
        assert CFG["chains_total"] % world == 0, "1024 chains do not divide over %d ranks" % world
        cfg["chains_per_gpu"] = CFG["chains_total"] // world
    scaling = "weak" if (args.weak or args.chains > 0) else "strong"
    n, C = cfg["n"], cfg["chains_per_gpu"]
    d = synth.geospatial(n, seed=cfg["seed"])
    if args.dense_z:
        # a dense, well-conditioned Z (Householder reflector): ZL = Z L has no structural zeros
        rng = np.random.default_rng(cfg["seed"] + 1)
        v = rng.standard_normal(n); v /= np.linalg.norm(v)
        d["Z"] = np.asfortranarray(np.eye(n) - 2.0 * np.outer(v, v))
    stream = torch.cuda.current_stream().cuda_stream
    ctx = api.Context(d["cov"], d["data"], d["eff_range"], d["Z"], d["X"], d["y"], d["family"], d["link"],
                      device=local_rank, stream=stream, rank=rank, world=world, reduce=hook)
    collective = "none (single process)"
    if world > 1:
        collective = "torch.distributed all_reduce hook (%s)" % backend
        if hook is None:
            # the library's own RCCL communicator (csrc/comm.hip).  Whether it came up is decided TOGETHER: a rank
            # that failed while the others succeeded would leave the job split over two transports
            ok = 1
            try:
                gdist.init_native_rccl(ctx, rank, world)
            except Exception as e:
                ok = 0
                print("bench: rank %d: native RCCL communicator failed (%r)" % (rank, e), file=sys.stderr, flush=True)
            flag = torch.tensor([ok], dtype=torch.int32, device="cuda")
            tdist.all_reduce(flag, op=tdist.ReduceOp.MIN)
            if int(flag.item()) == 1:
                collective = "librccl ncclAllReduce on the library's stream (native, no host sync)"
            else:       # a different GPU collective, not a CPU path: the line says which one ran
                ctx.close()
                hook = gdist.make_reduce_hook()
                ctx = api.Context(d["cov"], d["data"], d["eff_range"], d["Z"], d["X"], d["y"], d["family"],
                                  d["link"], device=local_rank, stream=stream, rank=rank, world=world, reduce=hook)
    def run(iters):
        return ctx.mcml_full(d["start"], mcnr=True, m=C, maxiter=iters, warmup=cfg["hmc_warmup"], tol=0.0,
                             verbose=False, lambda_=cfg["lambda_"], maxsteps=cfg["max_steps"],
                             target_accept=cfg["target_accept"], seed=cfg["seed"], chains=C,
                             maxfun=cfg["theta_maxfun"])

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            tdist.barrier()
            torch.cuda.synchronize()

    if emu:
        # recording run: the very iterations that are timed below, every rank's candidate thetas evaluated here
        ctx.emulate_world(emu, 1)
        run(args.steps)
        # replay once untimed (first use of the single-candidate path: workspace, graph capture), then again timed
        ctx.emulate_world(emu, 2)
        run(args.steps)
        ctx.emulate_world(emu, 2)
    elif args.warmup > 0:
        ctx.profile(enable=True, reset=True)                 # the warm-up runs with the markers on, as the timed steps do
        run(args.warmup)
    shard0 = ctx.shard_stats()
    ctx.profile(enable=True, reset=True)
    api.phase_ms(enable=True, reset=True)                    # host wall-clock per phase (csrc/trace.h): two clock reads per phase
    barrier()
    t0 = time.perf_counter()
    res = run(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    phases = api.phase_ms(enable=False)
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        tdist.all_reduce(tt, op=tdist.ReduceOp.MAX)
        dt = float(tt.item())
    prof = ctx.profile(enable=False)
    # what was computed (outside the timed region): the fit after `steps` iterations and the theta-step's objective there
    fit = {"beta": [float(x) for x in res["beta"]], "theta": [float(x) for x in res["theta"]], "sigma": float(res["sigma"])}
    if not emu:
        fit["mvn_ll_at_theta"] = float(ctx.mvn_ll(res["theta"]))
    th = np.asarray(res["theta"])
    fit_ok = bool(np.all(np.isfinite(res["beta"])) and np.all(np.isfinite(th)) and np.isfinite(res["sigma"]) and
                  0.1 < th[0] < 0.6 and 0.03 < th[1] < 0.3 and 0.7 < res["sigma"] < 1.3 and 0.0 < res["beta"][0] < 2.0 and
                  np.isfinite(fit.get("mvn_ll_at_theta", 0.0)))
    if args.n == CFG["n"] and not args.dense_z:
        assert fit_ok, "the timed iterations left the band round the generating values (theta 0.25, 0.1; sigma 1; beta 1): %r" % (fit,)
    shard1 = ctx.shard_stats()
    shard = {k: shard1[k] - shard0[k] for k in shard1}
    assert res["iters"] == args.steps, "timed region ran %d iterations, not %d" % (res["iters"], args.steps)

    if rank == 0:
        launches = prof["fwd_n"] + prof["bwd_n"]
        gemm_s = (prof["fwd_ms"] + prof["bwd_ms"]) * 1e-3
        avg_s = gemm_s / max(1, launches)
        # flops the kernels EXECUTE: with Z = I, ZL is triangular and the banded kernel skips its
        # all-zero K tiles (dgemm_band.h), so this is about half the dense 2 n Q C of SURVEY 8(d)
        executed = prof["fwd_flops"] * prof["fwd_n"] + prof["bwd_flops"] * prof["bwd_n"]
        achieved = executed / gemm_s / 1e12 if launches else 0.0
        dense_equiv = prof["dense_flops"] * launches / gemm_s / 1e12 if launches else 0.0
        # HBM bytes per launch from the PMC passes (FETCH_SIZE x2 + WRITE_SIZE, MI355X_MICROARCH.md "HBM"): a
        # separate rocprofv3 --pmc run of this same command (scripts/pmc_traffic.py writes the json); quoted
        # only when it was taken on this workload, with the file it came from
        traffic = None; traffic_src = None
        pj = os.path.join(ROOT, "profiles", "r03_hbm_traffic.json")
        if not os.path.exists(pj):
            pj = os.path.join(ROOT, "profiles", "r02_hbm_traffic.json")
        if os.path.exists(pj):
            try:
                tj = json.load(open(pj))
                if tj.get("n") == n and tj.get("chains") == C and bool(tj.get("dense_z")) == bool(args.dense_z):
                    traffic = tj.get("hbm_bytes_per_launch_%s" % prof["operator"])
                    traffic_src = "profiles/%s (%s)" % (os.path.basename(pj), tj.get("build", "?"))
            except Exception:
                traffic = None
        line = {
            "metric": "MCML iters/sec x m chains (simlik evals/sec)",
            "value": args.steps * C * world / dt,
            "unit": "simlik evals/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "gaussian geospatial n=Q=%d fexp dense Sigma%s, MCNR, m=%d chains in total = %d per GPU "
                                   "(HMC warmup %d + 1 draw/chain, %d leapfrog steps), theta-step BOBYQA budget %d evals"
                                   % (n, " with a DENSE Z (not the metric's Z = I)" if args.dense_z else "", C * world, C,
                                      cfg["hmc_warmup"], cfg["max_steps"], cfg["theta_maxfun"]),
                       "n": n, "Q": n, "m_per_gpu": C, "m_total": C * world, "parallelism": "chains x%d" % world,
                       "collective": collective,
                       "theta_step": ("sharded over candidate thetas: %d rounds, %d evaluations on this rank of %d in all, "
                                      "%d all-gather(s) of the samples (%.1f MB each)"
                                      % (shard["theta_rounds"], shard["theta_evals_own"], shard["theta_evals_all"],
                                         shard["gathers"], 8e-6 * shard["gather_doubles"] / max(1, shard["gathers"])))
                                     if shard["theta_rounds"] else "sequential BOBYQA on this GPU, %d evaluations per step" % cfg["theta_maxfun"],
                       "accept_rate": res["accept_rate"], "leapfrog_steps_last_iter": res["leapfrog_total"],
                       "phases_ms_per_step": {k: v / args.steps for k, v in phases.items()},
                       "fit": fit, "fit_ok": fit_ok},
            "roofline": {"bound": "mfma",
                         "kernel": "dgemm_%s_kernel (HMC forward / backward n x Q x C product, FP64 MFMA)"
                                   % ("band" if prof["operator"] == "banded" else "dlds"),
                         "achieved": achieved, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / FP64_MFMA_PEAK_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
                         "launches": prof["fwd_n_all"] + prof["bwd_n_all"], "launches_timed": launches,
                         "avg_launch_ms": avg_s * 1e3,
                         "flops_per_launch_executed": executed / max(1, launches),
                         "flops_per_launch_dense_2nQC": prof["dense_flops"],
                         "dense_equivalent_tflops": dense_equiv,
                         "note": "achieved = flops actually executed / kernel time; the banded kernel skips the "
                                 "structurally zero K tiles of the triangular ZL (Z = I), dense_equivalent_tflops "
                                 "prices the same launches at SURVEY 8(d)'s dense 2nQC",
                         "gemm_share_of_step": avg_s * (prof["fwd_n_all"] + prof["bwd_n_all"]) / dt},
        }
        if emu:
            line["config"]["as_rank_of"] = emu
            line["config"]["note"] = ("ONE rank of the %d-GPU strong-scaling job timed on one GPU: %d of the 1024 chains, this rank's "
                                      "share of every theta-step round; peers emulated as copies of this rank, the collectives' own "
                                      "time (1 all-gather of 41 MB + ~%d small all-reduces per step over xGMI) is not included. "
                                      "value = this rank's chains / its step time; x%d = the job's rate if every rank takes this long"
                                      % (emu, C, 1 + shard["theta_rounds"] // max(1, args.steps), emu))
            line["config"]["projected_job_value"] = args.steps * C * emu / dt
        if world == 1 and not args.no_cpu_baseline and not emu:
            try:
                cfg["chains_total_run"] = C * world; cfg["dense_z"] = bool(args.dense_z)
                line["cpu_baseline"] = cpu_baseline(d, cfg)
            except Exception as e:           # the baseline never blocks the GPU number
                line["cpu_baseline"] = {"value": None, "unit": "simlik evals/s", "cores": 0, "kind": "port",
                                        "sample": "failed: %r" % (e,)}
        if world == 1 and not emu and not args.no_other_configs and args.n == CFG["n"] and args.chains <= 0 and not args.dense_z:
            ctx.close()
            line["other_configs"] = other_configs(api, synth, stream)
        print(json.dumps(line), flush=True)
    ctx.close()
    if world > 1:
        tdist.destroy_process_group()


if __name__ == "__main__":
    main()
