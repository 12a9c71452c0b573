#!/usr/bin/env python3
"""Rays traced per bounce of one bench step (C3 by default): renders with max_depth = 1..D and differences the counters."""
import dataclasses, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401  (runtime ordering)
from cadrays_amd import scenes
from cadrays_amd.view import View
cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
sc = scenes.baseline_config(cfg)
D = sc.params.max_depth
v = View(0).load_scene(sc)
prev = 0
for d in range(1, D + 1):
    v.ChangeRenderingParams(max_depth=d)
    v.render(4); st = v.stats()
    tot = st["rays_nearest"] * 8          # 32 spp
    print(f"bounce {d-1}: {(tot - prev)/1e6:8.2f} M nearest rays per 32-spp step")
    prev = tot
