#!/usr/bin/env python3
"""Latency of crh_set_transforms -- the call the reference's manipulator makes on every frame while an object is dragged
(ImRaytraceControls.cxx:58-89, DataNode.cxx:239-242) -- on C3's million triangles grouped into G^3 objects.

  python tools/bench_transforms.py [G ...]            default: 10 30  (1 000 and 27 000 objects)

Reports the host-side time of one call (top-level tree rebuilt on the host, its nodes and the instance table copied
stream-ordered; nothing waits for the device), the time until the device has the new tree (call + crh_sync), and the rate of
the interactive loop `set_transforms; Redraw` at 1 spp.  Also the cost of the other per-frame setters."""
import dataclasses, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
from cadrays_amd import scenes
from cadrays_amd.view import View

Gs = [int(x) for x in sys.argv[1:]] or [10, 30]
sc = scenes.baseline_config("C3")
cen = sc.pos.reshape(-1, 3, 3).mean(1)
out = []
for G in Gs:
    cell = np.clip(((cen + 1.0) * 0.5 * G).astype(np.int32), 0, G - 1)
    obj = (cell[:, 0] * G + cell[:, 1]) * G + cell[:, 2]
    ids, inv = np.unique(obj, return_inverse=True)
    xf = np.tile(np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], np.float32), (len(ids), 1))
    two = dataclasses.replace(sc, tri_object=inv.astype(np.int32), obj_xform=xf)
    v = View(0).load_scene(two)
    v.render(1); v.sync()
    r = np.random.default_rng(1)
    moves = []
    for k in range(40):
        m = xf.copy(); m[:, [3, 7, 11]] += (r.random((len(ids), 3)).astype(np.float32) - 0.5) * 0.01     # every object nudged
        moves.append(m)
    for m in moves[:4]:
        v.set_transforms(m)
    v.sync()
    t_call, t_dev = [], []
    for m in moves[4:24]:
        t0 = time.perf_counter(); v.set_transforms(m); t1 = time.perf_counter(); v.sync(); t2 = time.perf_counter()
        t_call.append(t1 - t0); t_dev.append(t2 - t0)
    t0 = time.perf_counter()
    for m in moves[24:]:
        v.set_transforms(m); v.Redraw()
    v.sync()
    loop = (time.perf_counter() - t0) / len(moves[24:])
    mats = list(two.materials)
    t0 = time.perf_counter()
    for _ in range(50):
        v.set_materials(mats)
    t_mat = (time.perf_counter() - t0) / 50
    v.sync()
    t0 = time.perf_counter()
    for _ in range(50):
        v.set_lights(two.lights)
    t_lig = (time.perf_counter() - t0) / 50
    rec = {"objects": int(len(ids)), "set_transforms_call_ms_median": round(float(np.median(t_call)) * 1e3, 3),
           "set_transforms_call_ms_max": round(float(np.max(t_call)) * 1e3, 3),
           "set_transforms_until_on_device_ms_median": round(float(np.median(t_dev)) * 1e3, 3),
           "move_and_redraw_per_s_1spp": round(1.0 / loop, 1), "set_materials_call_ms": round(t_mat * 1e3, 4), "set_lights_call_ms": round(t_lig * 1e3, 4)}
    print(json.dumps(rec), flush=True)
    out.append(rec)
    v.close()
