"""Sustained shader clock and executed TFLOP/s of the banded FP64 MFMA kernel (debug hook)."""
import ctypes as C, sys
sys.path.insert(0, '.')
import numpy as np
from glmmrmcml_amd import _lib
L = _lib.lib()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
modes = [int(a) for a in sys.argv[3:]] or [0]
for mode in modes:
  for iters in ((20, 200, 2000) if mode == 0 else (500,)):
    out = np.zeros(3)
    _lib.check(L.glmmr_mcml_dbg_band_clocks(M, N, iters, mode, out.ctypes.data_as(C.POINTER(C.c_double))))
    print(f"mode {mode} iters {iters}: {out[0]*1e3:.1f} us/launch  shader clock {out[1]:.0f} MHz  executed {out[2]:.1f} TFLOP/s "
          f"-> {out[2] / (78.6 * out[1] / 2400.0) :.3f} of the MFMA peak at that clock", flush=True)
