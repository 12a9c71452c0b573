"""cadrays_amd -- MI355X-native progressive path tracer behind the CADRays rendering boundary.

Only what the hot path needs: csrc/ (hand-written gfx950 kernels + the C ABI), the ctypes binding,
the host mirror of the reference's View / BSDF / light interface, scene inputs and tile sharding.
"""
from . import abi, materials, scenes  # noqa: F401
from .materials import BSDF, Fresnel  # noqa: F401
from .scenes import Camera, Light, Params, Scene  # noqa: F401


def View(*a, **k):
    from .view import View as _V
    return _V(*a, **k)
