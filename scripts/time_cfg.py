"""times the sampler + steps on the HBM-bound configurations (1, 4, 5 of BASELINE.json)"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from glmmrmcml_amd import api, synth
which = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
C = int(sys.argv[2]) if len(sys.argv) > 2 else 512
t0 = time.time()
if which == "cfg4": d = synth.stepped_wedge(40, 8, 50)
elif which == "cfg5": d = synth.longitudinal(2000, 10)
else: d = synth.cluster_rct(10, 5, 10)
print(which, "n", d["n"], "Q", d["Q"], "P", d["P"], "synth %.1fs" % (time.time() - t0), flush=True)
ctx = api.Context(d["cov"], d["data"], d["eff_range"], d["Z"], d["X"], d["y"], d["family"], d["link"])
t0 = time.time(); ctx.update_L(d["theta"]); print("update_L %.4fs" % (time.time() - t0), flush=True)
ctx.profile(enable=True, reset=True)
for warm in (5, 50):
    t0 = time.time()
    dg = ctx.hmc_sample(d["beta"], 1.0, warm, C, 0.5, 10, 0.9, seed=1, chains=C)
    dt = time.time() - t0
    print(f"hmc warm={warm} chains={C}: {dt:.3f}s leapfrog {dg['leapfrog_total']} {dg['leapfrog_total']/dt:.3e} chain-steps/s acc={dg['accept_rate']:.3f} e={dg['mean_e']:.4f}", flush=True)
pr = ctx.profile(enable=False)
n, Q = d["n"], d["Q"]
print("fwd avg us", pr["fwd_ms"] / max(1, pr["fwd_n"]) * 1e3, "bwd avg us", pr["bwd_ms"] / max(1, pr["bwd_n"]) * 1e3)
bytes_fwd = 8.0 * (2 * n * C)  # MU + S written; gathers come from cache
bytes_bwd = 8.0 * (n * C + 4 * Q * C)
print("algorithmic GB/s fwd %.0f bwd %.0f" % (bytes_fwd / (pr["fwd_ms"] / max(1, pr["fwd_n"]) * 1e-3) / 1e9, bytes_bwd / (pr["bwd_ms"] / max(1, pr["bwd_n"]) * 1e-3) / 1e9))
t0 = time.time(); v = ctx.mvn_ll(d["theta"]); print("mvn_ll %.4fs" % (time.time() - t0), v)
t0 = time.time(); v = ctx.mvn_ll(d["theta"] * 1.05); print("mvn_ll %.4fs" % (time.time() - t0), v)
t0 = time.time(); r = ctx.mcnr(d["beta"], 1.0); print("mcnr %.4fs" % (time.time() - t0))
t0 = time.time(); r = ctx.loglik(d["beta"], 1.0); print("loglik %.4fs" % (time.time() - t0), r)
t0 = time.time(); r = ctx.mcml_optim(d["start"], mcnr=True); print("mcml_optim %.3fs" % (time.time() - t0), r["theta"])
