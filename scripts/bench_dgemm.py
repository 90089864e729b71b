"""GEMM micro-benchmark through the debug hook.
usage: python scripts/bench_dgemm.py [M N K b_nmajor tile iters [lower_only beta]]   (no arguments: a fixed sweep)
tile: -1 auto / 0..15 register-staged variants (dgemm_mfma.h), 20..25 = dgemm_dl tiles 1..6"""
import ctypes as C, sys
sys.path.insert(0, '.')
from glmmrmcml_amd import _lib
L = _lib.lib()
ms = C.c_double()


def run(M, N, K, nm, t, iters=10, lower=0, beta=0.0):
    _lib.check(L.glmmr_mcml_dbg_dgemm_bench2(M, N, K, nm, iters, t, lower, C.c_double(beta), C.byref(ms)))
    fl = 2.0 * M * N * K * (0.5 if lower else 1.0)
    print(f"M{M} N{N} K{K} nmajor={nm} tile={t} lower={lower} beta={beta} {ms.value * 1e3:.1f} us  "
          f"{fl / ms.value / 1e9:.2f} TFLOP/s", flush=True)


if len(sys.argv) >= 6:
    M, N, K, nm, t = (int(a) for a in sys.argv[1:6])
    run(M, N, K, nm, t, int(sys.argv[6]) if len(sys.argv) > 6 else 10, int(sys.argv[7]) if len(sys.argv) > 7 else 0,
        float(sys.argv[8]) if len(sys.argv) > 8 else 0.0)
else:
    for (M, N, K) in [(5000, 1024, 5000), (2000, 256, 2000), (8192, 8192, 8192)]:
        for nm in (0, 1):
            for t in [0, 1, 4, 5, 3]:
                run(M, N, K, nm, t)
