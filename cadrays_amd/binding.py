"""Thin ctypes binding over a library that exports the cadrays_hip.h entry points under a prefix.

The product binds prefix 'crh_' on libcadrays_hip.so (cadrays_amd/view.py); the test suite binds
prefix 'orc_' on the CPU oracle with this same class, so the parity tests drive both sides through
the same calls with the same bytes.
"""
import ctypes as C
import os
import numpy as np

from . import abi
from .materials import pack_materials

_f32p = C.POINTER(C.c_float)
_i32p = C.POINTER(C.c_int32)
_u32p = C.POINTER(C.c_uint32)
_u8p = C.POINTER(C.c_uint8)


class BackendError(RuntimeError):
    pass


def _fp(a):
    return a.ctypes.data_as(_f32p) if a is not None else None


_live = None      # weak set of open contexts: closed by an atexit hook, i.e. BEFORE the interpreter tears modules (and with them the HIP
                  # runtime torch loaded) down in arbitrary order -- a context destroyed after that calls into a dead runtime


def _close_all():
    for b in list(_live or ()):
        try:
            b.close()
        except Exception:
            pass


class Backend:
    """One rendering context ( == one V3d_View on one GPU )."""

    def __init__(self, lib, prefix, create_args=()):
        global _live
        self._lib, self._p = lib, prefix
        f = self._fn("create")
        f.restype = C.c_void_p
        self._ctx = C.c_void_p(f(*create_args))
        if not self._ctx.value:
            raise BackendError(f"{prefix}create failed")
        if _live is None:
            import atexit, weakref
            _live = weakref.WeakSet()
            if not os.environ.get("CRH_NO_ATEXIT_CLOSE"): atexit.register(_close_all)      # the switch exists to measure what the hook prevents
        _live.add(self)
        self.width = self.height = 0
        self._fn("last_error").restype = C.c_char_p

    def _fn(self, name):
        return getattr(self._lib, self._p + name)

    def _call(self, name, *args):
        rc = self._fn(name)(self._ctx, *args)                  # (restype of a ctypes function defaults to c_int)
        if rc != 0:
            msg = self._fn("last_error")(self._ctx)
            raise BackendError(f"{self._p}{name} -> {rc}: {msg.decode() if msg else ''}")
        return rc

    def close(self):
        if self._ctx and self._ctx.value:
            f = self._fn("destroy")
            f.restype = None
            f(self._ctx)
            self._ctx = C.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- scene ---------------------------------------------------------------------------
    def set_geometry(self, pos, nrm, tri, uv=None, tri_object=None, obj_xform=None):
        pos = np.ascontiguousarray(pos, np.float32).reshape(-1, 3)
        nrm = np.ascontiguousarray(nrm, np.float32).reshape(-1, 3)
        tri = np.ascontiguousarray(tri, np.int32).reshape(-1, 4)
        uv = None if uv is None else np.ascontiguousarray(uv, np.float32)
        to = None if tri_object is None else np.ascontiguousarray(tri_object, np.int32)
        self._has_objects = to is not None
        xf = None if obj_xform is None else np.ascontiguousarray(obj_xform, np.float32).reshape(-1, 12)
        self._call("set_geometry", _fp(pos), _fp(nrm), _fp(uv), C.c_uint32(len(pos)),
                   tri.ctypes.data_as(_i32p), C.c_uint32(len(tri)),
                   to.ctypes.data_as(_i32p) if to is not None else None,
                   _fp(xf), C.c_uint32(0 if xf is None else len(xf)))

    def set_transforms(self, obj_xform):
        """new per-object 3x4 transforms of a two-level scene: only the top-level tree is rebuilt (ImRaytraceControls.cxx:58-89)"""
        xf = np.ascontiguousarray(obj_xform, np.float32).reshape(-1, 12)
        self._call("set_transforms", _fp(xf), C.c_uint32(len(xf)))

    def set_visibility(self, visible):
        """Display / Erase without a rebuild (DataNode.cxx:304-344): one flag per object"""
        v = np.ascontiguousarray(np.asarray(visible) != 0, np.uint8)
        self._call("set_visibility", v.ctypes.data_as(C.POINTER(C.c_uint8)), C.c_uint32(len(v)))

    def add_object(self, pos, nrm, tri, xform, uv=None):
        """a new object into the built scene (an instance until the next full build); returns its object index"""
        pos = np.ascontiguousarray(pos, np.float32).reshape(-1, 3)
        nrm = np.ascontiguousarray(nrm, np.float32).reshape(-1, 3)
        tri = np.ascontiguousarray(tri, np.int32).reshape(-1, 4)
        uv = None if uv is None else np.ascontiguousarray(uv, np.float32)
        xf = np.ascontiguousarray(xform, np.float32).reshape(12)
        out = C.c_uint32(0)
        self._call("add_object", _fp(pos), _fp(nrm), _fp(uv), C.c_uint32(len(pos)), tri.ctypes.data_as(_i32p), C.c_uint32(len(tri)), _fp(xf), C.byref(out))
        return int(out.value)

    def get_tlas(self):
        r, n, b = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
        self._call("get_tlas", C.byref(r), C.byref(n), C.byref(b))
        return {"root": r.value, "n_instances": n.value, "n_blas_nodes": b.value}

    def set_materials(self, bsdfs):
        arr = pack_materials(bsdfs)
        self._call("set_materials", arr, C.c_uint32(len(bsdfs)))

    def set_lights(self, lights):
        arr = (abi.crh_light * max(len(lights), 1))()
        for i, l in enumerate(lights):
            arr[i].vec[:] = [float(x) for x in l.vec]
            arr[i].is_point = 1.0 if l.is_point else 0.0
            arr[i].emission[:] = [float(np.float32(c) * np.float32(l.intensity)) for c in l.color]
            arr[i].smoothness = float(l.smoothness)
        self._call("set_lights", arr, C.c_uint32(len(lights)))

    def set_envmap(self, env):
        if env is None:
            self._call("set_envmap", None, C.c_uint32(0), C.c_uint32(0))
        else:
            env = np.ascontiguousarray(env, np.float32)
            h, w, _ = env.shape
            self._call("set_envmap", _fp(env), C.c_uint32(w), C.c_uint32(h))

    def set_texture(self, slot, image):
        if image is None:
            self._call("set_texture", C.c_uint32(slot), None, C.c_uint32(0), C.c_uint32(0), C.c_uint32(3))
        else:
            image = np.ascontiguousarray(image, np.float32)
            h, w, ch = image.shape                       # (H, W, 3) RGB or (H, W, 4) RGBA
            self._call("set_texture", C.c_uint32(slot), _fp(image), C.c_uint32(w), C.c_uint32(h), C.c_uint32(ch))

    def set_camera(self, cam):
        c = abi.crh_camera()
        c.eye[:] = [float(x) for x in cam.eye]
        c.dir[:] = [float(x) for x in cam.dir]
        c.up[:] = [float(x) for x in cam.up]
        c.fovy_deg, c.aspect = float(cam.fovy_deg), float(cam.aspect)
        c.is_ortho, c.ortho_scale = int(cam.is_ortho), float(cam.ortho_scale)
        c.aperture_radius, c.focal_dist = float(cam.aperture_radius), float(cam.focal_dist)
        self._call("set_camera", C.byref(c))

    def set_params(self, p):
        q = abi.crh_params()
        q.width, q.height, q.max_depth = int(p.width), int(p.height), int(p.max_depth)
        q.radiance_clamp, q.two_sided, q.coherent_rng = float(p.radiance_clamp), int(p.two_sided), int(p.coherent_rng)
        q.seed, q.tile_size, q.tonemap_mode = int(p.seed), int(p.tile_size), int(p.tonemap_mode)
        q.exposure, q.white_point = float(p.exposure), float(p.white_point)
        q.background[:] = [float(x) for x in p.background]
        q.env_as_background, q.scene_epsilon = int(p.env_as_background), float(p.scene_epsilon)
        q.russian_roulette = int(p.russian_roulette)
        self._call("set_params", C.byref(q))
        self.width, self.height, self.tile_size = int(p.width), int(p.height), int(p.tile_size)

    def set_spec(self, **switches):
        """crh_set_spec: flip switches of include/crh_spec.h (abi.SPEC_DEFAULTS names them all); the ones not named return to their defaults.
        An unknown name is an error, not a silently ignored switch.  Restarts accumulation."""
        unknown = sorted(set(switches) - set(abi.SPEC_DEFAULTS))
        if unknown:
            raise ValueError(f"set_spec: no such switch {unknown}; include/crh_spec.h has {sorted(abi.SPEC_DEFAULTS)}")
        vals = dict(abi.SPEC_DEFAULTS); vals.update(switches)
        sp = abi.crh_spec(size=C.sizeof(abi.crh_spec))
        for name, ctype in abi.crh_spec._fields_[1:]:
            setattr(sp, name, float(vals[name]) if ctype is C.c_float else int(vals[name]))
        self._call("set_spec", C.byref(sp))

    def get_spec(self):
        sp = abi.crh_spec(size=C.sizeof(abi.crh_spec))      # in / out: the caller's struct size, exactly that many bytes are written
        self._call("get_spec", C.byref(sp))
        return sp.as_dict()

    def spec_order_exact(self):
        """the build-time switch CRH_SPEC_ORDER_EXACT of the loaded library (crh_spec.h #4)"""
        f = self._fn("spec_order_exact"); f.restype = C.c_int
        return int(f())

    def spec_anyhit_slot_order(self):
        """the build-time switch CRH_SPEC_ANYHIT_SLOT_ORDER of the loaded library (crh_spec.h #8)"""
        f = self._fn("spec_anyhit_slot_order"); f.restype = C.c_int
        return int(f())

    def load_scene(self, scene, prebuilt=None):
        """prebuilt = (nodes, prim_order) of another context's tree over the same geometry (crh_build_prebuilt) instead of building one here"""
        self.set_geometry(scene.pos, scene.nrm, scene.tri, scene.uv, getattr(scene, "tri_object", None), getattr(scene, "obj_xform", None))
        self.set_materials(scene.materials)
        self.set_lights(scene.lights)
        self.set_envmap(scene.env)
        self.set_camera(scene.camera)
        self.set_params(scene.params)
        for slot, img in enumerate(getattr(scene, "textures", []) or []):
            self.set_texture(slot, img)
        if getattr(scene, "spec", None):
            self.set_spec(**scene.spec)
        if prebuilt is not None:
            self.build_prebuilt(*prebuilt)
        else:
            self.build()
        return self

    # -- rendering -----------------------------------------------------------------------
    def build(self):
        self._call("build")

    def build_prebuilt(self, nodes, prim_order):
        """crh_build_prebuilt: this context takes the tree another one built for the same geometry (export_tree)"""
        nodes = np.ascontiguousarray(nodes).view(np.float32).reshape(-1, abi.NODE_DWORDS)
        order = np.ascontiguousarray(prim_order, np.uint32)
        self._call("build_prebuilt", _fp(nodes), C.c_uint32(len(nodes)), order.ctypes.data_as(_u32p), C.c_uint32(len(order)))

    def export_tree(self):
        """(nodes, prim_order) for build_prebuilt of another context: the node array and the leaf order (dword 3 of the leaf-ordered triangle records)"""
        nodes, tris = self.get_bvh()
        return nodes, np.ascontiguousarray(tris[:, 3]).view(np.uint32).copy()

    def reset(self):
        self._call("reset")

    def render(self, n_iterations=1):
        self._call("render", C.c_uint32(n_iterations))

    def render_tiles(self, tile_ids, first_sample, n_samples):
        t = np.ascontiguousarray(tile_ids, np.uint32)
        self._call("render_tiles", t.ctypes.data_as(_u32p), C.c_uint32(len(t)),
                   C.c_uint32(first_sample), C.c_uint32(n_samples))

    def set_adaptive(self, on=True, tiles_per_iteration=128):
        """AdaptiveScreenSampling / NbRayTracingTiles (SettingsWidget.cxx:427-477); restarts accumulation."""
        self._call("set_adaptive", C.c_int(int(on)), C.c_uint32(int(tiles_per_iteration)))

    def set_show_tiles(self, on=True):
        """ShowSamplingTiles (SettingsWidget.cxx:443-449): outline the tiles of the last adaptive iteration in read_ldr()."""
        self._call("set_show_tiles", C.c_int(int(on)))

    def tile_stats(self):
        n = C.c_uint32(0)
        self._call("get_tile_stats", None, None, C.byref(n))
        err, cnt = np.empty(n.value, np.float32), np.empty(n.value, np.uint32)
        self._call("get_tile_stats", _fp(err), cnt.ctypes.data_as(_u32p), C.byref(n))
        return err, cnt

    def n_tiles(self):
        ts = self.tile_size or 32
        return ((self.width + ts - 1) // ts) * ((self.height + ts - 1) // ts)

    def read_hdr(self):
        out = np.empty((self.height, self.width, 3), np.float32)
        self._call("read_hdr", _fp(out))
        return out

    def read_ldr(self):
        out = np.empty((self.height, self.width, 3), np.uint8)
        self._call("read_ldr", out.ctypes.data_as(_u8p))
        return out

    def read_ldr_begin(self):
        """queue tone map + read-back of the frame as submitted so far; returns at once (crh_read_ldr_begin)"""
        self._call("read_ldr_begin")
        self._rb_shapes = getattr(self, "_rb_shapes", []) + [(self.height, self.width, 3)]

    def read_ldr_end(self, out=None):
        """the oldest begun read-back (crh_read_ldr_end).  `out`: a C-contiguous uint8 array of the frame's shape to fill instead of a fresh one -- what a GUI
        host does (one staging buffer for the texture upload, INTEGRATION.md `myLdr`): a fresh 6 MB array per 1080p frame is 1500 first-touch page faults"""
        pending = getattr(self, "_rb_shapes", None)
        shape = pending[0] if pending else (self.height, self.width, 3)
        if out is None:
            out = np.empty(shape, np.uint8)
        elif out.dtype != np.uint8 or out.shape != tuple(shape) or not out.flags["C_CONTIGUOUS"] or not out.flags["WRITEABLE"]:
            raise ValueError(f"read_ldr_end(out=...): need a writeable C-contiguous uint8 array of shape {tuple(shape)}")
        self._call("read_ldr_end", out.ctypes.data_as(_u8p))
        if pending: pending.pop(0)
        return out

    def read_hdr_begin(self):
        """queue the HDR read-back of the frame as submitted so far; returns at once (crh_read_hdr_begin)"""
        self._call("read_hdr_begin")
        self._rbh_shapes = getattr(self, "_rbh_shapes", []) + [(self.height, self.width, 3)]

    def read_hdr_end(self):
        shape = self._rbh_shapes.pop(0) if getattr(self, "_rbh_shapes", None) else (self.height, self.width, 3)
        out = np.empty(shape, np.float32)
        self._call("read_hdr_end", _fp(out))
        return out

    def stats(self):
        s = abi.crh_stats()
        self._call("get_stats", C.byref(s))
        return s.as_dict()

    # -- kernel-level --------------------------------------------------------------------
    def trace_nearest(self, rays):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
        out = np.empty((len(rays), 4), np.float32)
        self._call("trace_nearest", _fp(rays), C.c_uint32(len(rays)), _fp(out))
        return out

    def trace_any(self, rays):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
        out = np.empty(len(rays), np.uint32)
        self._call("trace_any", _fp(rays), C.c_uint32(len(rays)), out.ctypes.data_as(_u32p))
        return out

    def get_bvh(self):
        nn, nt = C.c_uint32(0), C.c_uint32(0)
        self._call("get_bvh", None, C.byref(nn), None, C.byref(nt))
        nodes = np.empty((nn.value, abi.NODE_DWORDS), np.float32)      # 64-B stride nodes (include/crh_bvh_format.h)
        tris = np.empty((nt.value, 12), np.float32)
        self._call("get_bvh", _fp(nodes), C.byref(nn), _fp(tris), C.byref(nt))
        return nodes, tris
