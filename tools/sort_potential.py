#!/usr/bin/env python3
"""What would ordering a bounce's ray queue by origin buy on an HBM-resident scene?  Secondary rays exactly as the pipeline
produces them (camera rays in the kernel's 8x8-block pixel order -> first hits -> one random direction per hit, queue order =
pixel order) traced by crh_bench_trace (a) in that order, (b) ordered by a Morton code of the origin, (c) shuffled."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
from cadrays_amd import scenes
from cadrays_amd.view import View
cfg = sys.argv[1] if len(sys.argv) > 1 else "C5"
sc = scenes.baseline_config(cfg)
v = View(0).load_scene(sc)
W, H = 2048, 1024
cam = sc.camera
fwd = np.array(cam.dir, np.float64); fwd /= np.linalg.norm(fwd)
right = np.cross(fwd, np.array(cam.up, np.float64)); right /= np.linalg.norm(right); up = np.cross(right, fwd)
th = np.tan(np.radians(cam.fovy_deg) / 2)
# pixel order of the kernels: 32x32 tiles, 8x8 blocks inside
ys, xs = np.mgrid[0:H, 0:W]
key = ((ys // 32) * (W // 32) + xs // 32) * 1024 + (((ys % 32) // 8) * 4 + (xs % 32) // 8) * 64 + (ys % 8) * 8 + xs % 8
order = np.argsort(key.ravel(), kind="stable")
px, py = xs.ravel()[order], ys.ravel()[order]
nx = (px + 0.5) / W * 2 - 1; ny = 1 - (py + 0.5) / H * 2
d = fwd + right * (nx * th * W / H)[:, None] + up * (ny * th)[:, None]; d /= np.linalg.norm(d, axis=1, keepdims=True)
n = len(d)
rays = np.zeros((n, 8), np.float32); rays[:, :3] = np.array(cam.eye, np.float32); rays[:, 3] = 1e30; rays[:, 4:7] = d
h = v.trace_nearest(rays)
hit = h[:, 3].view(np.int32) >= 0
p = rays[hit, :3] + rays[hit, 4:7] * h[hit, :1]
r = np.random.default_rng(0)
d2 = r.normal(size=(len(p), 3)); d2 /= np.linalg.norm(d2, axis=1, keepdims=True)
sec = np.zeros((len(p), 8), np.float32); sec[:, :3] = p + 1e-4 * d2; sec[:, 3] = 1e30; sec[:, 4:7] = d2
def morton(q, bits=7):
    g = np.clip(((q + 1.0) * 0.5 * (1 << bits)).astype(np.int64), 0, (1 << bits) - 1)
    code = np.zeros(len(q), np.int64)
    for b in range(bits):
        for a in range(3):
            code |= ((g[:, a] >> b) & 1) << (3 * b + a)
    return code
out = {"config": cfg, "secondary_rays": int(len(sec))}
for name, idx in (("pipeline_order", np.arange(len(sec))), ("morton_sorted", np.argsort(morton(sec[:, :3]), kind="stable")), ("shuffled", r.permutation(len(sec)))):
    ms = v.bench_trace(np.ascontiguousarray(sec[idx]), repeat=10)
    out[name + "_ms"] = round(ms, 3)
print(json.dumps(out))
