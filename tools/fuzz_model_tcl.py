#!/usr/bin/env python3
"""Hunt beside tests/test_scene_tcl.py::test_cpp_reader_understands_model_tcl_like_the_python_reader: random scenes exported the way
CADRays exports them (cadrays_amd.scene_tcl.write_scene: model.tcl + binary PLY + PNG / JPEG textures and environment), with random
materials (all four Fresnel models), lights, cameras, texture scales and vlocation lines, read back by the Python reader and by the
C++ reader (cadrays_amd/host/model_tcl.hpp via model_tcl_dump); prints the seeds whose .crhscene bytes differ.  CPU only.
    python tools/fuzz_model_tcl.py [first] [last]"""
import dataclasses, os, re, shutil, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from PIL import Image
from cadrays_amd import scene_io, scenes
from cadrays_amd.materials import BSDF, Fresnel
from cadrays_amd.scene_tcl import read_scene, write_scene

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
exe = os.path.join(ROOT, "cadrays_amd", "host", "model_tcl_dump")
a, b = (int(sys.argv[1]) if len(sys.argv) > 1 else 0), (int(sys.argv[2]) if len(sys.argv) > 2 else 200)


def rand_fresnel(r):
    k = r.integers(0, 4)
    if k == 0: return Fresnel.CreateConstant(float(r.random()))
    if k == 1: return Fresnel.CreateSchlick(tuple(float(x) for x in r.random(3)))
    if k == 2: return Fresnel.CreateConductor(float(r.random() * 3 + 0.1), float(r.random() * 4 + 0.1))
    return Fresnel.CreateDielectric(float(r.random() * 1.5 + 1.0))


bad = []
for seed in range(a, b):
    r = np.random.default_rng(seed)
    nm = int(r.integers(1, 6))
    pos, nrm, tri = scenes.gen_scene(int(r.integers(nm, 400)), seed, nm)
    mats = []
    for m in range(nm):
        base = BSDF.CreateDiffuse(float(r.random()))
        mats.append(dataclasses.replace(base, Kc=tuple(float(x) for x in r.random(4)), Kd=tuple(float(x) for x in r.random(3)), Ks=tuple(float(x) for x in r.random(4)),
                                        Kt=tuple(float(x) for x in r.random(3)), Le=tuple(float(x) for x in r.random(3) * (r.random() < 0.3)),
                                        Absorption=tuple(float(x) for x in r.random(4)), FresnelCoat=rand_fresnel(r), FresnelBase=rand_fresnel(r)))
    textures, uv = [], None
    if r.random() < 0.7:
        uv = (r.random((len(pos), 2)) * 3 - 1).astype(np.float32)
        for slot in range(int(r.integers(1, 3))):
            w, h = int(r.integers(1, 20)), int(r.integers(1, 20))
            t = r.random((h, w, 3)).astype(np.float32)
            if r.random() < 0.4: t = np.concatenate([t, (r.random((h, w, 1)) > 0.5).astype(np.float32)], 2)
            textures.append(t)
        for m in range(nm):
            if r.random() < 0.6:
                mats[m] = dataclasses.replace(mats[m], texture=int(r.integers(0, len(textures))),
                                              texture_scale=(1.0, 1.0) if r.random() < 0.5 else (float(np.float32(r.random() * 4 + 0.1)), float(np.float32(r.random() * 4 + 0.1))))
    lights = []
    for _ in range(int(r.integers(0, 4))):
        v = tuple(float(x) for x in r.normal(size=3))
        lights.append(scenes.Light.positional(v, float(r.random()), float(r.random() * 50), tuple(float(x) for x in r.random(3))) if r.random() < 0.5
                      else scenes.Light.directional(v, float(r.random() * 0.5), float(r.random() * 10), tuple(float(x) for x in r.random(3))))
    d = r.normal(size=3); d /= np.linalg.norm(d)
    up = np.cross(np.cross(d, r.normal(size=3)), d); up /= np.linalg.norm(up)
    cam = scenes.Camera(eye=tuple(float(x) for x in r.normal(size=3) * 3), dir=tuple(float(x) for x in d), up=tuple(float(x) for x in up), fovy_deg=float(r.random() * 80 + 20),
                        is_ortho=bool(r.random() < 0.3), ortho_scale=float(r.random() * 3 + 0.2))
    env = (r.random((int(r.integers(1, 12)), int(r.integers(2, 24)), 3)) * 0.95).astype(np.float32) if r.random() < 0.6 else None
    sc = scenes.Scene(pos, nrm, tri, mats, lights=lights, camera=cam, env=env, uv=uv, textures=textures or None,
                      params=dataclasses.replace(scenes.Params(), max_depth=int(r.integers(1, 12))))
    tmp = tempfile.mkdtemp()
    try:
        model = write_scene(sc, os.path.join(tmp, "export"))
        txt = open(model).read()
        if env is not None and r.random() < 0.5:                     # the reference's default environment is a JPEG
            e8 = (np.sqrt(np.clip(env, 0, 1)) * 255 + 0.5).astype(np.uint8)
            Image.fromarray(e8).save(os.path.join(tmp, "export", "textures", "env.jpg"), quality=int(r.integers(40, 100)), subsampling=int(r.integers(0, 3)), progressive=bool(r.integers(0, 2)))
            txt = txt.replace("vtextureenv on $Root/textures/env.png", "vtextureenv on $Root/textures/env.jpg")
        names = re.findall(r"vdisplay (\S+)", txt)
        for nme in names:
            if r.random() < 0.4:
                q = r.normal(size=4); q /= np.linalg.norm(q)
                if r.random() < 0.7: txt += "vlocation %s -rotation %r %r %r %r\n" % (nme, *(float(x) for x in q))
                if r.random() < 0.5: txt += f"vlocation {nme} -scale {float(r.random() * 2 + 0.2)!r}\n"
                txt += "vlocation %s -location %r %r %r\n" % (nme, *(float(x) for x in r.normal(size=3)))
        open(model, "w").write(txt)
        w, h = int(r.integers(8, 200)), int(r.integers(8, 200))
        py, bld = read_scene(model, w, h)
        pa, pb = os.path.join(tmp, "py.crhscene"), os.path.join(tmp, "cpp.crhscene")
        scene_io.save_scene(py, pa)
        p = subprocess.run([exe, model, pb, f"{w}x{h}"], capture_output=True, text=True)
        if p.returncode or "not honoured" in p.stderr or bld.unsupported or open(pa, "rb").read() != open(pb, "rb").read():
            bad.append(seed); print(seed, p.returncode, p.stderr.strip()[:200], bld.unsupported, flush=True)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
print(f"{b - a} exported scenes, mismatches:", bad)
