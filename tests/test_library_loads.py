"""CPU-side check that the C-ABI library loads and exports every symbol that
include/glmmr_mcml_c.h declares (no compute without a GPU)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "glmmr_mcml_c.h")).read()
    return sorted(set(re.findall(r"\b(glmmr_mcml_\w+)\s*\(", hdr)))


def test_exports_every_declared_symbol():
    from glmmrmcml_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    L = _lib.lib()
    syms = _declared_symbols()
    assert len(syms) >= 3
    missing = [s for s in syms if not hasattr(L, s)]
    assert not missing, missing


def test_compute_without_gpu_fails_loudly():
    import ctypes as C
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from glmmrmcml_amd import _lib
    L = _lib.lib()
    ms = C.c_double()
    rc = L.glmmr_mcml_dbg_dgemm_bench(64, 64, 64, 0, 1, -1, C.byref(ms))
    assert rc != 0
    with pytest.raises(_lib.McmlError):
        _lib.check(rc)


def test_phase_clock_is_host_only():
    """glmmr_mcml_dbg_phase_ms touches no device: enable / read / reset work on a host without a GPU and start at zero"""
    from glmmrmcml_amd import api
    api.phase_ms(enable=True, reset=True)
    got = api.phase_ms(enable=False)
    assert set(got) == {"sample", "beta_step", "theta_step", "refresh"} and all(v == 0.0 for v in got.values())
