import ctypes as C, sys
sys.path.insert(0, '.')
from glmmrmcml_amd import _lib
L = _lib.lib()
ms = C.c_double()
tile = int(sys.argv[1]) if len(sys.argv) > 1 else 0
nm = int(sys.argv[2]) if len(sys.argv) > 2 else 0
_lib.check(L.glmmr_mcml_dbg_dgemm_bench(5000, 1024, 5000, nm, 5, tile, C.byref(ms)))
print("tile", tile, "nmajor", nm, ms.value, "ms", 2.0 * 5000 * 1024 * 5000 / ms.value / 1e9, "TF", flush=True)
