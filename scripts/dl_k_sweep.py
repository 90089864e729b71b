"""the Cholesky's trailing update C -= A A' (lower tiles only, C read-modify-written) on the LDS-DMA kernel (dgemm_dl.h) as a
function of K: one pass with K = 1024 against eight passes with K = 128 over the same C
usage: python scripts/dl_k_sweep.py [M=4000]"""
import ctypes as C, sys
sys.path.insert(0, ".")
from glmmrmcml_amd import _lib
L = _lib.lib()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
ms = C.c_double()
for tile in (23, 24):               # dgemm_dl tiles 4 (128 x 128, 2-stage) and 5 (64 x 128, 3-stage)
    for K in (128, 256, 512, 1024):
        _lib.check(L.glmmr_mcml_dbg_dgemm_bench2(M, M, K, 1, 20, tile, 1, C.c_double(1.0), C.byref(ms)))
        fl = 2.0 * M * M / 2 * K
        print("tile %d  M=N=%d K=%4d lower-only: %.3f ms per launch = %.1f TFLOP/s; the K=1024 worth of updates: %.3f ms" % (tile - 19, M, K, ms.value, fl / ms.value / 1e9, ms.value * 1024 / K), flush=True)
