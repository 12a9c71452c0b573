#!/usr/bin/env python3
"""Latency of crh_set_visibility and crh_add_object -- Display / Erase of one object (the eye icons of the scene tree, `rtdisplay` / `rterase`:
DataNode.cxx:304-344, ImportExportPlugin.cxx:373-425) and `rtmeshread` into a running viewer -- on C3's million triangles grouped into G^3 objects,
beside what the same edit cost up to round 5: crh_set_geometry + crh_build of the whole scene.

  python tools/bench_visibility.py [G ...]            default: 3  (27 objects of ~37 000 triangles)

Per G: host time of one hide / one show call, time until the device holds the patched records (call + crh_sync), the first frame after the edit
(call + one Redraw + sync), the rate of the edited scene against the scene REBUILT without the object (wide batches, the bench's regime), whether the
two images are bit-identical, and the add-object cost for an object of the same size."""
import dataclasses, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
from cadrays_amd import scenes
from cadrays_amd.view import View

Gs = [int(x) for x in sys.argv[1:]] or [3]
sc = scenes.baseline_config("C3")
cen = sc.pos.reshape(-1, 3, 3).mean(1)
I12 = np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], np.float32)


def rate(v, spp=32, reps=2):
    v.reset(); v.render(spp); v.sync()
    st0 = v.stats(); t0 = time.perf_counter()
    for _ in range(reps):
        v.render(spp)
    v.sync()
    dt = time.perf_counter() - t0; st = v.stats()
    return ((st["rays_nearest"] + st["rays_any"]) - (st0["rays_nearest"] + st0["rays_any"])) / dt / 1e6


for G in Gs:
    cell = np.clip(((cen + 1.0) * 0.5 * G).astype(np.int32), 0, G - 1)
    obj = ((cell[:, 0] * G + cell[:, 1]) * G + cell[:, 2]).astype(np.int32)
    nO = G ** 3
    two = dataclasses.replace(sc, tri_object=obj, obj_xform=np.tile(I12, (nO, 1)))
    victim = (nO // 2)                                                      # the object in the middle of the cloud
    t0 = time.perf_counter(); v = View(0).load_scene(two); v.sync(); t_build = time.perf_counter() - t0
    v.render(1); v.sync()
    lone = []
    for _ in range(8):                                                       # the lone frame of the untouched scene, for scale
        v.reset(); v.sync(); t0 = time.perf_counter(); v.Redraw(); v.sync(); lone.append(time.perf_counter() - t0)
    vis = np.ones(nO, np.uint8)
    calls = {"hide": [], "show": []}; dev = {"hide": [], "show": []}; frame = {"hide": [], "show": []}
    for rep in range(6):
        for what, flag in (("hide", 0), ("show", 1)):
            vis[victim] = flag
            t0 = time.perf_counter(); v.set_visibility(vis); t1 = time.perf_counter(); v.sync(); t2 = time.perf_counter()
            v.Redraw(); v.sync(); t3 = time.perf_counter()
            if rep:                                                          # the first pair allocates the patch staging of this object
                calls[what].append(t1 - t0); dev[what].append(t2 - t0); frame[what].append(t3 - t0)
    vis[victim] = 0; v.set_visibility(vis)
    r_hidden = rate(v)
    v.reset(); v.render(4); hid = v.read_hdr()
    keep = obj != victim
    t0 = time.perf_counter()
    w = View(0).load_scene(dataclasses.replace(two, tri=two.tri[keep], tri_object=obj[keep])); w.sync()
    t_rebuild = time.perf_counter() - t0
    r_rebuilt = rate(w)
    w.reset(); w.render(4); same = bool(np.array_equal(hid.view(np.uint32), w.read_hdr().view(np.uint32)))
    # the same object handed over as a NEW one into the scene built without it
    t = two.tri[~keep]; vid, inv = np.unique(t[:, :3], return_inverse=True)
    tri1 = np.concatenate([inv.reshape(-1, 3).astype(np.int32), t[:, 3:4]], 1)
    t0 = time.perf_counter(); w.add_object(two.pos[vid], two.nrm[vid], tri1, I12); t1 = time.perf_counter(); w.sync(); t2 = time.perf_counter()
    add_call_ms, add_dev_ms = (t1 - t0) * 1e3, (t2 - t0) * 1e3
    r_added = rate(w)
    med = lambda a: round(float(np.median(a)) * 1e3, 3)
    lone_hidden = []
    for _ in range(8):
        v.reset(); v.sync(); t0 = time.perf_counter(); v.Redraw(); v.sync(); lone_hidden.append(time.perf_counter() - t0)
    print(json.dumps({"objects": nO, "lone_frame_ms_untouched": round(float(np.median(lone[1:])) * 1e3, 3), "lone_frame_ms_with_the_object_hidden": round(float(np.median(lone_hidden[1:])) * 1e3, 3), "triangles_of_the_object": int((~keep).sum()), "scene_hand_over_and_build_s": round(t_build, 3),
                      "rebuild_without_the_object_s": round(t_rebuild, 3),
                      "hide_call_ms": med(calls["hide"]), "show_call_ms": med(calls["show"]),
                      "hide_until_on_device_ms": med(dev["hide"]), "show_until_on_device_ms": med(dev["show"]),
                      "hide_plus_first_frame_ms": med(frame["hide"]), "show_plus_first_frame_ms": med(frame["show"]),
                      "mrays_hidden": round(r_hidden, 1), "mrays_rebuilt_without_it": round(r_rebuilt, 1), "hidden_over_rebuilt": round(r_hidden / r_rebuilt, 4),
                      "image_bit_identical_to_rebuilt": same,
                      "add_object_call_ms": round(add_call_ms, 2), "add_object_until_on_device_ms": round(add_dev_ms, 2),
                      "mrays_with_added_instance": round(r_added, 1)}), flush=True)
    v.close(); w.close()
