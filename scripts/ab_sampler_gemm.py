"""HIP-event time of the sampler's two products (forward / backward, GEMM + fix-up) at a few chain counts:
usage: python scripts/ab_sampler_gemm.py [n=5000] [chains ...]   (A/B: run twice with GLMMR_MCML_BAND_FIXUP=0|1 etc.)"""
import sys
sys.path.insert(0, '.')
from glmmrmcml_amd import api, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
Cs = [int(x) for x in sys.argv[2:]] or [1024, 128]
d = synth.geospatial(n)
ctx = api.Context(d["cov"], d["data"], d["eff_range"], d["Z"], d["X"], d["y"], d["family"], d["link"])
ctx.update_L(d["theta"])
for C in Cs:
    ctx.hmc_sample(d["beta"], d["sigma"], 2, C, 5.0, 10, 0.9, seed=1, chains=C)
    ctx.profile(enable=True, reset=True)
    dg = ctx.hmc_sample(d["beta"], d["sigma"], 20, C, 5.0, 10, 0.9, seed=1, chains=C)
    p = ctx.profile(enable=False)
    print("n=%d C=%d fwd %.1f us  bwd %.1f us (timed launches %d) accept %.3f" % (n, C, 1e3 * p["fwd_ms"] / max(1, p["fwd_n"]), 1e3 * p["bwd_ms"] / max(1, p["bwd_n"]), p["fwd_n"], dg["accept_rate"]), flush=True)
