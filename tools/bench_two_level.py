#!/usr/bin/env python3
"""Throughput of the TWO-LEVEL traversal kernels -- what a real CADRays scene uses: every displayed AIS object is an instance with its
own location (ImRaytraceControls.cxx:85-89, DataNode.cxx:239-242) -- against the single-level kernels the BASELINE configs run.
C3's million triangles grouped into G^3 objects (spatial cells; translated or rotated placements), same camera, 32 spp batches.

  python tools/bench_two_level.py [G ...] [identity|translated|rotated ...]          default: 1 4 10 (1, 64, 1000 objects), all three placements
"""
import dataclasses, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
from cadrays_amd import scenes
from cadrays_amd.view import View

KINDS = [k for k in sys.argv[1:] if not k.isdigit() and not k.startswith("spp")] or ["identity", "translated", "rotated"]
Gs = [int(x) for x in sys.argv[1:] if x.isdigit()] or [1, 4, 10]
SPP = next((int(k[3:]) for k in sys.argv[1:] if k.startswith("spp")), 32)      # samples per batch: `spp128` = the batch bench.py times
sc = scenes.baseline_config("C3")
cen = sc.pos.reshape(-1, 3, 3).mean(1)


def rate(v):
    v.render(SPP); v.sync(); v.reset()
    s0 = v.stats(); t0 = time.perf_counter()
    for _ in range(3):
        v.render(SPP)
    v.sync(); dt = time.perf_counter() - t0; s1 = v.stats()
    rays = (s1["rays_nearest"] + s1["rays_any"]) - (s0["rays_nearest"] + s0["rays_any"])
    return rays / dt * 1e-6


v = View(0).load_scene(sc)
print(json.dumps({"scene": "single level", "mrays_per_s": round(rate(v), 1)}), flush=True)
v.close()
for G in Gs:
    cell = np.clip(((cen + 1.0) * 0.5 * G).astype(np.int32), 0, G - 1)
    obj = (cell[:, 0] * G + cell[:, 1]) * G + cell[:, 2]
    ids, inv = np.unique(obj, return_inverse=True)
    for kind in KINDS:
        xf = np.tile(np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], np.float32), (len(ids), 1))
        pos = sc.pos.copy().reshape(-1, 3, 3); nrm = sc.nrm.copy().reshape(-1, 3, 3)
        if kind != "identity":
            r = np.random.default_rng(G)
            t = (r.random((len(ids), 3)).astype(np.float32) - 0.5) * 0.5
            if kind == "rotated":                                   # object space = world rotated about z by a per-object angle, then shifted
                a = r.random(len(ids)) * 2 * np.pi; c, s = np.cos(a).astype(np.float32), np.sin(a).astype(np.float32)
                R = np.zeros((len(ids), 3, 3), np.float32); R[:, 0, 0] = c; R[:, 0, 1] = -s; R[:, 1, 0] = s; R[:, 1, 1] = c; R[:, 2, 2] = 1
            else:
                R = np.tile(np.eye(3, dtype=np.float32), (len(ids), 1, 1))
            # world = R p_obj + t  =>  p_obj = R^T (p_world - t)
            Rt = np.transpose(R, (0, 2, 1))
            pos = np.einsum("tij,tvj->tvi", Rt[inv], pos - t[inv][:, None, :]).astype(np.float32)
            nrm = np.einsum("tij,tvj->tvi", Rt[inv], nrm).astype(np.float32)
            xf = np.concatenate([R, t[:, :, None]], 2).reshape(len(ids), 12).astype(np.float32)
        two = dataclasses.replace(sc, pos=pos.reshape(-1, 3), nrm=nrm.reshape(-1, 3), tri_object=inv.astype(np.int32), obj_xform=xf)
        v = View(0).load_scene(two)                                  # placed at build time: baked into one tree (round 3)
        print(json.dumps({"scene": f"two-level, {len(ids)} objects, {kind}, as built (baked)", "mrays_per_s": round(rate(v), 1), "instances": v.get_tlas()["n_instances"]}), flush=True)
        v.close()
        if kind != "identity":                                       # the same world, every object an INSTANCE: built at the identity, then all moved into place
            ident_all = np.tile(np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], np.float32), (len(ids), 1))
            v = View(0).load_scene(dataclasses.replace(two, obj_xform=ident_all)); v.set_transforms(xf)
            print(json.dumps({"scene": f"two-level, {len(ids)} objects, {kind}, every object an instance", "mrays_per_s": round(rate(v), 1), "instances": v.get_tlas()["n_instances"]}), flush=True)
            v.close()

# ---- static / moved split: what a CADRays session does -- the scene is loaded with every object where its vertices say, then the gizmo drags ONE
# object (src/ImGui/ImRaytraceControls.cxx:64,88).  1000 objects at the identity; 1, then 10 of them translated a little: throughput against the flat
# rate, latency of the first move (the object's tree is built, its triangles in the static tree disabled) and of the following ones.
if "split" in KINDS or len(sys.argv) == 1:
    print(json.dumps({"spp_per_batch": SPP}), flush=True)
    G = 10
    cell = np.clip(((cen + 1.0) * 0.5 * G).astype(np.int32), 0, G - 1)
    obj = (cell[:, 0] * G + cell[:, 1]) * G + cell[:, 2]
    ids, inv = np.unique(obj, return_inverse=True)
    ident = np.tile(np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], np.float32), (len(ids), 1))
    v = View(0).load_scene(dataclasses.replace(sc, tri_object=inv.astype(np.int32), obj_xform=ident))
    flat = rate(v)
    print(json.dumps({"scene": f"split: {len(ids)} objects, none moved", "mrays_per_s": round(flat, 1)}), flush=True)
    for _ in range(3):                                               # an interactive session has been issuing Redraw()s all along: the small-batch
        v.Redraw()                                                   # streams exist (their one-time creation is not part of a move)
    v.sync(); v.reset()
    r = np.random.default_rng(7)
    picks = r.permutation(len(ids))[:10]
    for n_moved in (1, 10):
        xf = ident.copy()
        lat = []
        for k in picks[:n_moved]:
            xf[k, 3::4] = (r.random(3).astype(np.float32) - 0.5) * 0.05
            v.sync(); t0 = time.perf_counter(); v.set_transforms(xf); t1 = time.perf_counter(); v.Redraw(); v.sync(); t2 = time.perf_counter()
            lat.append((t1 - t0, t2 - t0))
        again = []
        for _ in range(20):                                          # the drag goes on: the same objects, new offsets
            for k in picks[:n_moved]:
                xf[k, 3::4] += np.float32(0.001)
            v.sync(); t0 = time.perf_counter(); v.set_transforms(xf); again.append(time.perf_counter() - t0)
        m = rate(v)
        print(json.dumps({"scene": f"split: {len(ids)} objects, {n_moved} moved", "mrays_per_s": round(m, 1), "vs_none_moved": round(m / flat, 4),
                          "first_move_call_ms": [round(a * 1e3, 3) for a, _ in lat][-3:], "first_move_until_frame_ms": [round(b * 1e3, 3) for _, b in lat][-3:],
                          "next_moves_call_ms_median": round(float(np.median(again)) * 1e3, 3), "instances": v.get_tlas()["n_instances"]}), flush=True)
    v.set_transforms(ident)
    print(json.dumps({"scene": "split: everything back at the identity", "mrays_per_s": round(rate(v), 1), "instances": v.get_tlas()["n_instances"]}), flush=True)
    v.close()
