#!/usr/bin/env python3
"""C3's million-triangle scene rendered as a TWO-LEVEL scene: triangles grouped into G^3 spatial cells = objects with identity
transforms (what a CAD assembly of many parts looks like to the backend).  Prints Mrays/s next to the single-level figure."""
import dataclasses, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
from cadrays_amd import scenes
from cadrays_amd.view import View

G = int(sys.argv[1]) if len(sys.argv) > 1 else 10
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 32
sc = scenes.baseline_config("C3")
cen = sc.pos.reshape(-1, 3, 3).mean(1)                              # the generator emits 3 private vertices per triangle
cell = np.clip(((cen + 1.0) * 0.5 * G).astype(np.int32), 0, G - 1)
obj = (cell[:, 0] * G + cell[:, 1]) * G + cell[:, 2]
ids, inv = np.unique(obj, return_inverse=True)
xf = np.tile(np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], np.float32), (len(ids), 1))
two = dataclasses.replace(sc, tri_object=inv.astype(np.int32), obj_xform=xf)
for name, s in (("single-level", sc), (f"two-level, {len(ids)} objects", two)):
    v = View(0).load_scene(s)
    v.render(spp); v.sync(); v.reset()
    t = time.perf_counter(); v.render(spp); v.sync(); dt = time.perf_counter() - t
    st = v.stats()
    print(f"{name:32s} {st['rays_nearest'] / dt / 1e6:8.1f} Mrays/s   {dt * 1e3:7.1f} ms for {spp} spp")
