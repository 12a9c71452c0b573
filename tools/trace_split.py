"""C3 in 1000 objects, 128-spp batches: two batches with nothing moved, then `CRH_MOVED` (default 1) objects dragged and two more -- the workload whose
per-kernel times say where a split scene loses against the flat one:
rocprofv3 --kernel-trace --output-format csv -d <dir> -o t -- python3 tools/trace_split.py"""
import dataclasses, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa
from cadrays_amd import scenes
from cadrays_amd.view import View
sc = scenes.baseline_config("C3")
cen = sc.pos.reshape(-1, 3, 3).mean(1); G = 10
cell = np.clip(((cen + 1.0) * 0.5 * G).astype(np.int32), 0, G - 1)
ids, inv = np.unique((cell[:, 0] * G + cell[:, 1]) * G + cell[:, 2], return_inverse=True)
ident = np.tile(np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], np.float32), (len(ids), 1))
v = View(0).load_scene(dataclasses.replace(sc, tri_object=inv.astype(np.int32), obj_xform=ident))
for _ in range(3): v.render(128)
v.sync(); print("MARK moved", flush=True)
r = np.random.default_rng(7); xf = ident.copy()
for k in r.permutation(len(ids))[:int(os.environ.get("CRH_MOVED", "1"))]:
    xf[k, 3::4] = (r.random(3).astype(np.float32) - 0.5) * 0.05
v.set_transforms(xf)
for _ in range(3): v.render(128)
v.sync()
