#!/usr/bin/env python3
"""N lone frames (crh_reset + crh_render(1) + crh_sync each) of a BASELINE config: the workload profiles/frame_profile.sh puts under rocprofv3.
   python tools/lone_frames.py [--config C3] [--frames 12]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ap = argparse.ArgumentParser(); ap.add_argument("--config", default="C3"); ap.add_argument("--frames", type=int, default=12)
a = ap.parse_args()
import torch  # noqa: F401
from cadrays_amd import scenes
from cadrays_amd.view import View
v = View(0).load_scene(scenes.baseline_config(a.config))
ts = []
for _ in range(a.frames):
    v.reset(); v.sync()
    t = time.perf_counter(); v.Redraw(); v.sync(); ts.append((time.perf_counter() - t) * 1e3)
print("lone frame ms:", " ".join("%.2f" % x for x in ts))
