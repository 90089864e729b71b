import ctypes as C, sys
sys.path.insert(0, '.')
import numpy as np
from glmmrmcml_amd import api, synth, _lib
d = synth.geospatial(300)
ctx = api.Context(d["cov"], d["data"], d["eff_range"])
out = (C.c_ulonglong * 10)()
_lib.check(_lib.lib().glmmr_mcml_dbg_leaf_profile(ctx._h, out))
t = [int(x) for x in out]
names = ["load", None, None, None, None, "write L", "inv diag", "inv rows", "write Linv"]
print("total cycles", t[9] - t[0], "= %.1f us at 2.4 GHz" % ((t[9] - t[0]) / 2400.0))
print("(P1+sync) total ", t[2], " P2", t[3], " P3 (wave 0)", t[4])
print("P1 wave 0: loads", t[6], " chain", t[7], " stores", t[8], " wait at sync", t[1])
