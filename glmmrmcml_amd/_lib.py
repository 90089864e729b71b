"""Loads the in-tree HIP library.  There is no CPU fallback: if the library is
missing, or no MI355X is visible when a compute entry point is called, the call
fails loudly."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# GLMMR_MCML_LIB: developer override for A/B timing of two builds of the same library
LIB_PATH = os.environ.get("GLMMR_MCML_LIB") or os.path.join(_HERE, "libglmmr_mcml_hip.so")
_lib = None


class McmlError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("glmmr_mcml error %d: %s" % (code, msg))
        self.code = code


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "%s not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950); glmmrmcml_amd has no CPU fallback" % LIB_PATH)
        _lib = C.CDLL(LIB_PATH)
        _lib.glmmr_mcml_last_error.restype = C.c_char_p
    return _lib


def check(rc):
    if rc != 0:
        raise McmlError(rc, lib().glmmr_mcml_last_error().decode(errors="replace"))
