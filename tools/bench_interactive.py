#!/usr/bin/env python3
"""The reference's interactive regime: one Redraw() = +1 sample per pixel per GUI frame (AppViewer.cxx:1045-1047), on a BASELINE
config at its full resolution.  Prints Redraw/s without look-ahead (crh_set_lookahead(1)) -- free-running, with an LDR read-back
after every frame (what a GUI that shows every frame does), and with speculative look-ahead 16 for comparison.

  python tools/bench_interactive.py [--config C3] [--frames 64]        (CRH_LANES=1 disables the multi-stream small-batch path)"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
from cadrays_amd import scenes
from cadrays_amd.view import View

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="C3")
ap.add_argument("--frames", type=int, default=64)
a = ap.parse_args()
sc = scenes.baseline_config(a.config)
v = View(0).load_scene(sc)
out = {"config": a.config, "lanes": os.environ.get("CRH_LANES", "default"), "lane_grid_trace": os.environ.get("CRH_LANE_GRID_TRACE", "default")}
for name, k, readback in (("free_running", 1, False), ("with_ldr_readback", 1, True), ("with_async_ldr_readback", 1, "async"), ("lookahead16_free_running", 16, False)):
    v.set_lookahead(k); v.reset()
    for _ in range(8 if k == 1 else 2 * k):
        v.Redraw()
    v.sync()
    n = a.frames if k == 1 else max(a.frames, 4 * k)
    t0 = time.perf_counter()
    for i in range(n):
        v.Redraw()
        if readback == "async":                                     # every frame is displayed, two frames later: the read-backs of frames i - 1 and i stay in flight
            if i >= 2: v.read_ldr_end()
            v.read_ldr_begin()
        elif readback:
            v.read_ldr()
    if readback == "async": v.read_ldr_end(); v.read_ldr_end()
    v.sync()
    dt = time.perf_counter() - t0
    st = v.stats()
    out[name + "_redraw_per_s"] = round(n / dt, 1)
# adaptive screen sampling (SettingsWidget.cxx:427-477): NbRayTracingTiles tiles per iteration, drawn on the device; beside it a
# plain render of as many fixed tiles per call (equal paths per iteration)
import numpy as np
for ntiles in (128, 512):
    v.set_lookahead(1); v.set_adaptive(True, ntiles)
    for _ in range(16):
        v.Redraw()
    v.sync()
    t0 = time.perf_counter()
    for _ in range(4 * a.frames):
        v.Redraw()
    v.sync()
    out[f"adaptive_{ntiles}_tiles_redraw_per_s"] = round(4 * a.frames / (time.perf_counter() - t0), 1)
    v.set_adaptive(False, ntiles)
    tiles = np.random.default_rng(1).permutation(v.n_tiles())[:ntiles].astype(np.uint32)
    for i in range(16):
        v.render_tiles(tiles, i, 1)
    v.sync()
    t0 = time.perf_counter()
    for i in range(4 * a.frames):
        v.render_tiles(tiles, 16 + i, 1)
    v.sync()
    out[f"fixed_{ntiles}_tiles_calls_per_s"] = round(4 * a.frames / (time.perf_counter() - t0), 1)
print(json.dumps(out), flush=True)
