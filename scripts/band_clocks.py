"""standalone timing of the banded FP64 MFMA kernel (plain store epilogue): ms per launch, shader MHz, executed TFLOP/s
usage: python scripts/band_clocks.py [M=5000] [N=1024] [iters=50] [mode=0]"""
import ctypes as C, sys
sys.path.insert(0, '.')
import numpy as np
from glmmrmcml_amd import _lib
M = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
it = int(sys.argv[3]) if len(sys.argv) > 3 else 50
mode = int(sys.argv[4]) if len(sys.argv) > 4 else 0
out = np.zeros(3)
_lib.check(_lib.lib().glmmr_mcml_dbg_band_clocks(M, N, it, mode, out.ctypes.data_as(C.POINTER(C.c_double))))
print("band kernel M=%d N=%d mode=%d: %.1f us per launch, %.0f MHz, %.2f executed TFLOP/s" % (M, N, mode, out[0] * 1e3, out[1], out[2]))
