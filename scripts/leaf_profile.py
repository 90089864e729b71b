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
print("load          ", t[1] - t[0])
print("(a) diag tiles ", t[2])
print("(b) panel solve", t[3])
print("(c) trailing   ", t[4])
print("factor total   ", t[5] - t[1])
print("write L        ", t[6] - t[5])
print("last inverse row", t[7] - t[6], " (rows 0..6 run under phase (a) of the following step)")
print("write Linv     ", t[9] - t[8])
