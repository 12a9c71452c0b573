#!/usr/bin/env python3
"""Per-tile cost of a frame on ONE GPU, and what it predicts for the tile-sharded multi-GPU render (the 8-GPU node is
the driver's, this pool has 1-GPU boxes): every 32 x 32 tile is rendered on its own with the visit counters on; cost = inner-node visits +
triangle tests (what the traversal kernels' time follows).  For N = 2 / 4 / 8 and each tile -> GPU assignment the shard cost is summed and
    imbalance = max shard cost / mean shard cost        (the slowest GPU sets the frame time: predicted scaling efficiency = 1 / imbalance)
is printed: `t mod N` over the row-major numbering (rounds 1-2), the Z-order (Morton) interleave of cadrays_amd/sharding.py, and contiguous
blocks of rows (what NOT to do).

  python tools/tile_cost.py [--config C3] [--spp 4] [--out profiles/r3/tile_cost_C3.json]
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
from cadrays_amd import scenes, sharding
from cadrays_amd.view import View

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="C3"); ap.add_argument("--spp", type=int, default=4); ap.add_argument("--out", default="")
a = ap.parse_args()
sc = scenes.baseline_config(a.config)
v = View(0).load_scene(sc)
v.enable_counters(True); v.reset()
nt = v.n_tiles(); tx = sharding.tiles_x_of(v)
cost = np.zeros(nt); rays = np.zeros(nt)
prev = v.stats()
t0 = time.time()
for t in range(nt):
    v.render_tiles(np.array([t], np.uint32), 0, a.spp)
    s = v.stats()
    cost[t] = (s["nodes_nearest"] + s["tris_nearest"] + s["nodes_any"] + s["tris_any"]) - (prev["nodes_nearest"] + prev["tris_nearest"] + prev["nodes_any"] + prev["tris_any"])
    rays[t] = (s["rays_nearest"] + s["rays_any"]) - (prev["rays_nearest"] + prev["rays_any"])
    prev = s
out = {"config": a.config, "spp": a.spp, "tiles": int(nt), "tiles_x": int(tx), "seconds": round(time.time() - t0, 1),
       "tile_cost": {"mean": float(cost.mean()), "min": float(cost.min()), "max": float(cost.max()), "cv": float(cost.std() / cost.mean())},
       "tile_rays": {"mean": float(rays.mean()), "min": float(rays.min()), "max": float(rays.max())}, "imbalance": {}}
ty = nt // tx
for n in (2, 4, 8):
    row = {}
    row["t mod N (row-major)"] = [cost[r::n].sum() for r in range(n)]
    row["morton interleave (cadrays_amd/sharding.py)"] = [cost[sharding.tiles_for_rank(nt, r, n, tx)].sum() for r in range(n)]
    rows = np.array_split(np.arange(ty), n)
    row["contiguous rows"] = [cost.reshape(ty, tx)[rr].sum() for rr in rows]
    out["imbalance"][str(n)] = {k: {"max_over_mean": round(float(max(vv) / np.mean(vv)), 4), "predicted_efficiency": round(float(np.mean(vv) / max(vv)), 4)} for k, vv in row.items()}
print(json.dumps(out, indent=1))
if a.out:
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(out, open(a.out, "w"), indent=1)
