"""cadrays_amd -- MI355X-native progressive path tracer behind the CADRays rendering boundary.

Only what the hot path needs: csrc/ (hand-written gfx950 kernels + the C ABI), the ctypes binding,
the host mirror of the reference's View / BSDF / light interface, scene inputs and tile sharding.
"""


def pipeline_capacity():
    """(max frames in flight, hardware queues) of free-running Redraw()s in THIS process: crh_query_pipeline_capacity.  The HIP runtime maps streams onto
    GPU_MAX_HW_QUEUES hardware queues (default 4) and reads that variable at the process's first HIP call; the library reads it at its first use and
    never writes the environment -- a host that wants eight frames in flight (455 instead of 400 Redraw()/s on the 1 M-triangle scene) exports
    GPU_MAX_HW_QUEUES=16 itself before anything touches the GPU (bench.py, tests/conftest.py and the headless host do)."""
    import ctypes as C
    from ._lib import load_library
    lib = load_library()
    frames, queues = C.c_uint32(0), C.c_int(0)
    lib.crh_query_pipeline_capacity(C.byref(frames), C.byref(queues))
    return int(frames.value), int(queues.value)


from . import abi, materials, scenes  # noqa: E402,F401
from .materials import BSDF, Fresnel  # noqa: E402,F401
from .scenes import Camera, Light, Params, Scene  # noqa: E402,F401


def View(*a, **k):
    from .view import View as _V
    return _V(*a, **k)
