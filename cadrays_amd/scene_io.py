"""Flat binary scene file (.crhscene) for the headless C++ driver (cadrays_amd/host/cadrays_headless.cpp).
Layout: magic 'CRHS', version u32 (= 2), counts {nV, nT, nM, nL, envW, envH} u32, crh_camera, crh_params, then
pos[3nV] nrm[3nV] f32, tri[4nT] i32, crh_bsdf[nM], crh_light[nL], env[3*envW*envH] f32; version 2 appends
{has_uv, nO, nTex} u32, uv[2nV] f32 if has_uv, tri_object[nT] i32 + obj_xform[12nO] f32 if nO > 0 (two-level scene),
and per texture slot {w, h, channels} u32 + texels[w*h*channels] f32 (w = 0: empty slot).
(The reference's own on-disk scene format, model.tcl + PLY, is the 'next' row of SURVEY.md section 8f.)"""
import ctypes as C
import struct

import numpy as np

from . import abi
from .materials import pack_materials


def save_scene(scene, path):
    cam, par = abi.crh_camera(), abi.crh_params()
    c, p = scene.camera, scene.params
    cam.eye[:] = [float(x) for x in c.eye]; cam.dir[:] = [float(x) for x in c.dir]; cam.up[:] = [float(x) for x in c.up]
    cam.fovy_deg, cam.aspect, cam.is_ortho = float(c.fovy_deg), float(c.aspect), int(c.is_ortho)
    cam.ortho_scale, cam.aperture_radius, cam.focal_dist = float(c.ortho_scale), float(c.aperture_radius), float(c.focal_dist)
    par.width, par.height, par.max_depth = int(p.width), int(p.height), int(p.max_depth)
    par.radiance_clamp, par.two_sided, par.coherent_rng = float(p.radiance_clamp), int(p.two_sided), int(p.coherent_rng)
    par.seed, par.tile_size, par.tonemap_mode = int(p.seed), int(p.tile_size), int(p.tonemap_mode)
    par.exposure, par.white_point = float(p.exposure), float(p.white_point)
    par.background[:] = [float(x) for x in p.background]
    par.env_as_background, par.scene_epsilon, par.russian_roulette = int(p.env_as_background), float(p.scene_epsilon), int(p.russian_roulette)
    lights = (abi.crh_light * max(len(scene.lights), 1))()
    for i, l in enumerate(scene.lights):
        lights[i].vec[:] = [float(x) for x in l.vec]
        lights[i].is_point = 1.0 if l.is_point else 0.0
        lights[i].emission[:] = [float(np.float32(x) * np.float32(l.intensity)) for x in l.color]
        lights[i].smoothness = float(l.smoothness)
    env = scene.env
    eh, ew = (env.shape[0], env.shape[1]) if env is not None else (0, 0)
    with open(path, "wb") as f:
        f.write(b"CRHS")
        f.write(struct.pack("<7I", 2, len(scene.pos), len(scene.tri), len(scene.materials), len(scene.lights), ew, eh))
        f.write(bytes(cam)); f.write(bytes(par))
        f.write(np.ascontiguousarray(scene.pos, np.float32).tobytes())
        f.write(np.ascontiguousarray(scene.nrm, np.float32).tobytes())
        f.write(np.ascontiguousarray(scene.tri, np.int32).tobytes())
        f.write(bytes(pack_materials(scene.materials))[:C.sizeof(abi.crh_bsdf) * len(scene.materials)])
        f.write(bytes(lights)[:C.sizeof(abi.crh_light) * len(scene.lights)])
        if env is not None:
            f.write(np.ascontiguousarray(env, np.float32).tobytes())
        uv, tobj, xf = scene.uv, getattr(scene, "tri_object", None), getattr(scene, "obj_xform", None)
        textures = list(getattr(scene, "textures", None) or [])
        n_obj = len(xf) if (tobj is not None and xf is not None) else 0
        f.write(struct.pack("<3I", int(uv is not None), n_obj, len(textures)))
        if uv is not None:
            f.write(np.ascontiguousarray(uv, np.float32).tobytes())
        if n_obj:
            f.write(np.ascontiguousarray(tobj, np.int32).tobytes())
            f.write(np.ascontiguousarray(xf, np.float32).reshape(-1).tobytes())
        for t in textures:
            if t is None:
                f.write(struct.pack("<3I", 0, 0, 0))
            else:
                t = np.ascontiguousarray(t, np.float32)
                f.write(struct.pack("<3I", t.shape[1], t.shape[0], t.shape[2])); f.write(t.tobytes())
    return path
