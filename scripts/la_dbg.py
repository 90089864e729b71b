import sys
sys.path.insert(0, '.')
import numpy as np
from glmmrmcml_amd import api, synth
d = synth.cluster_rct(ncl=6, nt=3, nind=8, family="poisson")
with api.Context(d["cov"], d["data"], d["eff_range"], d["Z"], d["X"], d["y"], d["family"], d["link"]) as ctx:
    try:
        r = ctx.mcml_la(d["start"], nr=True, maxiter=6, verbose=True, trace=1)
        print(r)
    except Exception as e:
        print("ERR", e)
