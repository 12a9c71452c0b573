#!/usr/bin/env python3
"""How long does one traversal launch take as a function of its ray count?  (The floor of a small launch decides the
interactive regime)  Rays: origins on random triangles of C3's scene, uniform directions."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
from cadrays_amd import scenes
from cadrays_amd.view import View
sc = scenes.baseline_config("C3")
v = View(0).load_scene(sc)
r = np.random.default_rng(0)
N = 1 << 21
t = r.integers(0, len(sc.tri), N)
org = sc.pos[sc.tri[t, 0]] + np.float32(1e-4)
d = r.normal(size=(N, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
rays = np.zeros((N, 8), np.float32); rays[:, :3] = org; rays[:, 3] = 1e30; rays[:, 4:7] = d
for n in (64, 256, 1024, 4096, 16384, 65536, 262144, 1048576, 2097152):
    ms = v.bench_trace(rays[:n], repeat=20)
    print(json.dumps({"rays": n, "launch_us": round(ms * 1e3, 1), "mrays_per_s": round(n / ms / 1e3, 1)}), flush=True)
