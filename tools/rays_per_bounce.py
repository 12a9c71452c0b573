#!/usr/bin/env python3
"""Rays, node visits and triangle tests per bounce of the bench workload (C3 by default): renders with max_depth = 1..D
with the counters on and differences them."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401  (runtime ordering)
from cadrays_amd import scenes
from cadrays_amd.view import View
cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
sc = scenes.baseline_config(cfg)
D = sc.params.max_depth
v = View(0).load_scene(sc)
v.enable_counters(True)
prev = dict(rays_nearest=0, nodes_nearest=0, tris_nearest=0)
for d in range(1, D + 1):
    v.ChangeRenderingParams(max_depth=d)
    v.render(4); st = v.stats()
    dr = st["rays_nearest"] - prev["rays_nearest"]
    print(f"bounce {d-1}: {dr * 8 / 1e6:8.2f} M rays per 32 spp   {(st['nodes_nearest'] - prev['nodes_nearest']) / max(dr, 1):6.2f} node visits/ray   "
          f"{(st['tris_nearest'] - prev['tris_nearest']) / max(dr, 1):5.2f} triangle tests/ray")
    prev = st
