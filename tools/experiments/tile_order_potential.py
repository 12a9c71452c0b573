#!/usr/bin/env python3
"""Experiment (round 6): would the ORDER in which a lone frame's tiles are claimed shorten its drain?  The frame kernel hands path slots out tile by tile from one
cursor; whatever is claimed last runs out its bounces on an emptying chip (profiles/r6/lone_frame.md 1b: the drain is 0.43 of a lone frame).  Pixels do not depend on
the order (the RNG is seeded per pixel), so the tile list of crh_render_tiles may be any permutation.  Here: the per-tile ray counts of one counted frame, then lone
frames through crh_render_tiles with the tiles in natural order, most expensive first, cheapest first, centre-out, and at random.
   python tools/experiments/tile_order_potential.py [--config CAD1M]"""
import argparse, json, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np
import torch  # noqa: F401
from cadrays_amd import scenes, sharding
from cadrays_amd.view import View

ap = argparse.ArgumentParser(); ap.add_argument("--config", default="CAD1M"); ap.add_argument("--trials", type=int, default=15)
a = ap.parse_args()
sc = scenes.baseline_config(a.config)
v = View(0).load_scene(sc)
nt = v.n_tiles(); tx = sharding.tiles_x_of(v); ty = (nt + tx - 1) // tx
v.enable_counters(True); v.reset()
rays = np.zeros(nt); prev = v.stats()
for t in range(nt):
    v.render_tiles(np.array([t], np.uint32), 0, 1)
    s = v.stats(); rays[t] = (s["rays_nearest"] + s["rays_any"]) - (prev["rays_nearest"] + prev["rays_any"]); prev = s
v.enable_counters(False); v.reset()
ids = np.arange(nt, dtype=np.uint32)
cx, cy = (tx - 1) / 2.0, (ty - 1) / 2.0
centre = np.argsort(np.hypot((ids % tx) - cx, (ids // tx) - cy), kind="stable").astype(np.uint32)
orders = {"natural (0 .. n-1)": ids, "most rays first": np.argsort(-rays, kind="stable").astype(np.uint32), "fewest rays first": np.argsort(rays, kind="stable").astype(np.uint32),
          "centre out": centre, "random": np.random.default_rng(1).permutation(nt).astype(np.uint32)}
for _ in range(40):                                            # the library's feeder tuning first
    if not v.frame_tuning()["enabled"] or v.frame_tuning()["feeders"]: break
    v.reset(); v.Redraw(); v.sync()
out = {"config": a.config, "tiles": int(nt), "rays_per_tile": {"mean": float(rays.mean()), "min": float(rays.min()), "max": float(rays.max()), "cv": float(rays.std() / rays.mean())},
       "frame_tuning": v.frame_tuning(), "lone_frame_ms": {}}
ref = None
for rep in range(2):
    for name, order in orders.items():
        ts = []
        for _ in range(a.trials):
            v.reset(); v.sync()
            t0 = time.perf_counter(); v.render_tiles(order, 0, 1); v.sync(); ts.append((time.perf_counter() - t0) * 1e3)
        out["lone_frame_ms"].setdefault(name, []).append(round(statistics.median(ts[2:]), 3))
        img = v.read_hdr()
        if ref is None: ref = img
        assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), name      # the image does not depend on the order
ts = []
for _ in range(a.trials):
    v.reset(); v.sync(); t0 = time.perf_counter(); v.Redraw(); v.sync(); ts.append((time.perf_counter() - t0) * 1e3)
out["lone_frame_ms"]["Redraw() (the library's own tile list)"] = [round(statistics.median(ts[2:]), 3)]
order, n_re = v.tile_order()
q = nt // 4
out["library_order"] = {"reorders": n_re, "mean_rays_first_quarter": float(rays[order[:q]].mean()), "mean_rays_last_quarter": float(rays[order[-q:]].mean()),
                        "same_as_row_major": bool(np.array_equal(order, ids))}
ts = []
for _ in range(a.trials):
    v.reset(); v.sync(); t0 = time.perf_counter(); v.render_tiles(order, 0, 1); v.sync(); ts.append((time.perf_counter() - t0) * 1e3)
out["lone_frame_ms"]["crh_render_tiles with the library's list"] = [round(statistics.median(ts[2:]), 3)]
print(json.dumps(out))
