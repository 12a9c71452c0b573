"""Loader of the C-ABI HIP module.  There is no fallback: a missing or unloadable
libcadrays_hip.so is an error, never a silent CPU path."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CRH_LIB_PATH") or os.path.join(_HERE, "libcadrays_hip.so")   # CRH_LIB_PATH: A/B builds of the same ABI
_LIB = None


class HipModuleMissing(RuntimeError):
    pass


def load_library():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise HipModuleMissing(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). cadrays_amd has no CPU rendering path.")
        try:
            _LIB = C.CDLL(LIB_PATH)
        except OSError as e:
            raise HipModuleMissing(f"cannot load {LIB_PATH}: {e}") from e
    return _LIB
