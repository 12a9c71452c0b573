"""Host-side mirror of the rendering boundary CADRays drives (reference file:line in brackets).

    view = View(device=0)                       # new OpenGl_GraphicDriver + V3d_Viewer + CreateView   [AppViewer.cxx:601-638]
    view.load_scene(scene)                      # AIS Display + SetBSDF + lights + env + camera          [AisMesh.cxx:357-423, MaterialEditor.cxx:331-337]
    view.ChangeRenderingParams(max_depth=5)     # Graphic3d_RenderingParams field writes                 [SettingsWidget.cxx:263-477]
    view.Redraw()                               # +1 sample per pixel                                    [AppViewer.cxx:1047]
    img = view.BufferDump(BT_RGB_RayTraceHdrLeft)   # linear HDR read-back                               [AppGui.cxx:345-349]

Everything below the method names is libcadrays_hip.so (hand-written gfx950 kernels) through ctypes.
"""
import ctypes as C
import dataclasses

import numpy as np

from . import abi
from ._lib import load_library
from .binding import Backend, BackendError

BT_RGB = "Graphic3d_BT_RGB"                             # AppViewer.cxx:1259-1261
BT_RGB_RayTraceHdrLeft = "Graphic3d_BT_RGB_RayTraceHdrLeft"   # AppGui.cxx:349


class View(Backend):
    def __init__(self, device=0):
        super().__init__(load_library(), "crh_", (C.c_int(int(device)),))
        self.device = int(device)
        self._params = None
        import os
        from . import pipeline_capacity
        frames, _ = pipeline_capacity()
        if frames > 3 and "CRH_PIPE_DEPTH" not in os.environ:      # the process has the hardware queues for a deeper frame pipeline (crh_query_pipeline_capacity)
            self.set_pipeline_depth(frames)

    # ---- V3d_View vocabulary ----------------------------------------------------------------
    def Redraw(self):
        """One progressive iteration: +1 spp over the whole target (AppViewer.cxx:1047)."""
        self.render(1)

    def BufferDump(self, buffer_type=BT_RGB):
        if buffer_type == BT_RGB_RayTraceHdrLeft:
            return self.read_hdr()
        if buffer_type == BT_RGB:
            return self.read_ldr()
        raise ValueError(buffer_type)

    def set_params(self, p):
        self._params = dataclasses.replace(p)
        super().set_params(p)

    def ChangeRenderingParams(self, **fields):
        """Write Graphic3d_RenderingParams fields; like OCCT, any change restarts accumulation."""
        if self._params is None:
            raise RuntimeError("set_params / load_scene first")
        self.set_params(dataclasses.replace(self._params, **fields))

    # ---- device-side extras -----------------------------------------------------------------
    def sync(self):
        self._call("sync")

    def set_lookahead(self, frames):
        """trace `frames` Redraw()s ahead in one wide batch; images after every Redraw stay bit-identical"""
        self._call("set_lookahead", C.c_uint32(int(frames)))

    def set_lookahead_auto(self, max_frames):
        """crh_set_lookahead_auto: one sample right after a restart, then batches of 4, 16, ... max_frames; 0 / 1 = off"""
        self._call("set_lookahead_auto", C.c_uint32(int(max_frames)))

    def set_pipeline_depth(self, frames):
        """crh_set_pipeline_depth: frames in flight of free-running Redraw()s, 2 .. cadrays_amd.pipeline_capacity()[0] (3 on the runtime's default four hardware queues)"""
        self._call("set_pipeline_depth", C.c_uint32(int(frames)))

    def set_path_budget(self, max_paths):
        """crh_set_path_budget: at most this many path slots (196 B each) in flight per batch; images do not depend on it"""
        self._call("set_path_budget", C.c_uint64(int(max_paths)))

    def get_path_budget(self):
        n = C.c_uint64(0)
        self._call("get_path_budget", C.byref(n))
        return int(n.value)

    def frame_tuning(self):
        """crh_get_frame_tuning: the frame kernel's feeder count chosen by measurement after every build (no image depends on it)"""
        out = (C.c_uint32 * 5)()
        self._call("get_frame_tuning", out)
        return {"enabled": bool(out[0]), "feeders": int(out[1]), "frames_measured": int(out[2]), "mean_us_3_feeders": int(out[3]), "mean_us_4_feeders": int(out[4])}

    def tile_order(self):
        """crh_get_tile_order: the order crh_render lists the tiles in (no pixel depends on it), and how often it has been replaced"""
        n, r = C.c_uint32(0), (C.c_uint64 * 7)()
        self._call("get_tile_order", None, C.byref(n), r)
        order = np.zeros(n.value, np.uint32)
        self._call("get_tile_order", order.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(n), r)
        self.tile_order_calls = {"sorted": int(r[1]), "row_major": int(r[2]), "frames_collected": int(r[3]), "verdict": int(r[4]), "mean_us_sorted": int(r[5]), "mean_us_row_major": int(r[6])}
        return order, int(r[0])

    def packet_stats(self):
        """camera rays walked as packets since the last restart, and those of them handed to the per-ray fall-back pass (ties at equal distance)"""
        a, b = C.c_uint64(0), C.c_uint64(0)
        self._call("get_packet_stats", C.byref(a), C.byref(b))
        return {"packet_rays": int(a.value), "fallback_rays": int(b.value)}

    def set_schedule(self, mode):
        """crh_set_schedule: abi.SCHEDULE_AUTO / _WIDE (the big-batch schedule bench.py times) / _SMALL; images do not depend on it"""
        self._call("set_schedule", C.c_int(int(mode)))

    def enable_counters(self, on=True):
        self._call("enable_counters", C.c_int(int(on)))

    def enable_kernel_timing(self, on=True):
        self._call("enable_kernel_timing", C.c_int(int(on)))

    def kernel_timing(self):
        ms, n, allms = C.c_double(0), C.c_uint64(0), C.c_double(0)
        self._call("get_kernel_timing", C.byref(ms), C.byref(n), C.byref(allms))
        return {"trace_nearest_ms_total": ms.value, "trace_nearest_launches": n.value, "render_ms_total": allms.value}

    def save_accum(self):
        """checkpoint: (H, W, 4) float32 accumulator (rgb mean + sample count) and the iteration counter"""
        out = np.empty((self.height, self.width, 4), np.float32)
        n = C.c_uint32(0)
        self._call("save_accum", out.ctypes.data_as(C.POINTER(C.c_float)), C.byref(n))
        return out, n.value

    def load_accum(self, rgba, frames_done):
        rgba = np.ascontiguousarray(rgba, np.float32)
        assert rgba.shape == (self.height, self.width, 4)
        self._call("load_accum", rgba.ctypes.data_as(C.POINTER(C.c_float)), C.c_uint32(int(frames_done)))

    def scene_bytes(self):
        """HBM residency of the built scene: 64-B-stride nodes, 64-B-stride triangle records, 64-B shading records -- and, for a scene without placed
        objects, the 128-B packet nodes the camera rays of wide batches read (kernels.hip k_expand_packet_nodes; not part of what the per-ray walk touches)"""
        nn, nt = C.c_uint32(0), C.c_uint32(0)
        self._call("get_bvh", None, C.byref(nn), None, C.byref(nt))
        two_level = getattr(self, "_has_objects", False)
        return {"nodes": 4 * abi.NODE_DWORDS * nn.value, "triangles": 64 * nt.value, "shading": 64 * nt.value,
                "packet_nodes": 0 if two_level else 128 * nn.value, "n_nodes": nn.value, "n_triangles": nt.value}

    def accum_device_ptr(self):
        p, n = C.c_void_p(0), C.c_uint64(0)
        self._call("accum_device_ptr", C.byref(p), C.byref(n))
        return p.value, n.value

    @staticmethod
    def reduce(views, root=0, fake_devices=False):
        """crh_reduce: assemble the tile-sharded frame of `views` (one context per GPU) on views[root]; that view's
        read_hdr / read_ldr / save_accum return the assembled frame until it renders again.  fake_devices: the test hook
        crh_debug_reduce_fake_devices (the RCCL branch on contexts that share a device)"""
        lib = views[root]._lib
        arr = (C.c_void_p * len(views))(*[v._ctx.value for v in views])
        fn = lib.crh_debug_reduce_fake_devices if fake_devices else lib.crh_reduce
        fn.restype = C.c_int
        rc = fn(arr, C.c_uint32(len(views)), C.c_uint32(int(root)))
        if rc != 0:
            msg = views[root]._fn("last_error")(views[root]._ctx)
            raise BackendError(f"crh_reduce -> {rc}: {msg.decode() if msg else ''}")

    def bench_trace(self, rays, any_hit=False, repeat=10):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
        ms = C.c_float(0)
        self._call("bench_trace", rays.ctypes.data_as(C.POINTER(C.c_float)), C.c_uint32(len(rays)), C.c_int(int(any_hit)),
                   C.c_uint32(repeat), C.byref(ms))
        return ms.value

    def debug_math(self, fn, a, b=None):
        a = np.ascontiguousarray(a, np.float32)
        b = np.ascontiguousarray(b if b is not None else np.zeros_like(a), np.float32)
        out, out2 = np.empty_like(a), np.empty_like(a)
        fp = C.POINTER(C.c_float)
        self._call("debug_math", C.c_int(fn), a.ctypes.data_as(fp), b.ctypes.data_as(fp), out.ctypes.data_as(fp),
                   out2.ctypes.data_as(fp), C.c_uint32(a.size))
        return out, out2


    def debug_bsdf(self, fn, bsdf, a, b=None, two_sided=True):
        """crh_debug_bsdf: fn 0 eval (f cos), 1 pdf, 2 sample, 3 Fresnel of the coat on the device; a, b: (n, 3) float32"""
        a = np.ascontiguousarray(a, np.float32).reshape(-1, 3)
        b = np.ascontiguousarray(b if b is not None else np.zeros_like(a), np.float32).reshape(-1, 3)
        n = len(a)
        out = np.empty((n, {0: 3, 1: 1, 2: 8, 3: 3}[fn]), np.float32)
        m = bsdf.to_abi()
        fp = C.POINTER(C.c_float)
        self._call("debug_bsdf", C.c_int(fn), C.byref(m), a.ctypes.data_as(fp), b.ctypes.data_as(fp), out.ctypes.data_as(fp),
                   C.c_uint32(n), C.c_int(int(two_sided)))
        return out


def build_bvh_host(pos, tri, threads=0):
    """Run the product's BVH builder on the host only (no GPU): returns (nodes[n,16] u32-as-f32, prim_order[nT] u32)."""
    lib = load_library()
    pos = np.ascontiguousarray(pos, np.float32).reshape(-1, 3)
    tri = np.ascontiguousarray(tri, np.int32).reshape(-1, 4)
    nn = C.c_uint32(0)
    fp, ip, up = C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_uint32)
    args = (pos.ctypes.data_as(fp), C.c_uint32(len(pos)), tri.ctypes.data_as(ip), C.c_uint32(len(tri)), C.c_int(threads))
    nodes = np.empty((max(len(tri), 4), abi.NODE_DWORDS), np.float32)   # nodes <= max(1, ~nT/2), 64 B each
    order = np.empty(len(tri), np.uint32)
    rc = lib.crh_build_bvh_host(*args, nodes.ctypes.data_as(fp), C.byref(nn), order.ctypes.data_as(up))
    if rc != 0:
        raise RuntimeError(f"crh_build_bvh_host -> {rc}")
    return nodes[:nn.value].copy(), order
