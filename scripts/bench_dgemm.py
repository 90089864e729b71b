import ctypes as C, sys
sys.path.insert(0, '.')
from glmmrmcml_amd import _lib
L = _lib.lib()
ms = C.c_double()
for (M,N,K) in [(5000,1024,5000),(2000,256,2000),(8192,8192,8192)]:
    for nm in (0,1):
        for t in [0,1,4,5,3]:
            _lib.check(L.glmmr_mcml_dbg_dgemm_bench(M,N,K,nm,10,t,C.byref(ms)))
            print(f"M{M} N{N} K{K} nmajor={nm} tile={t} {ms.value:.3f} ms  {2.0*M*N*K/ms.value/1e9:.2f} TFLOP/s", flush=True)
