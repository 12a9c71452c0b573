#!/usr/bin/env python3
"""What the frame kernel's engines did in one lone frame, from an instrumented build (tools/ab_build.sh fstats "-DCRH_FRAME_STATS=1"):
   CRH_LIB_PATH=cadrays_amd/variants/fstats.so python tools/frame_stats.py [--config C3]"""
import argparse, ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ap = argparse.ArgumentParser(); ap.add_argument("--config", default="C3")
a = ap.parse_args()
import torch  # noqa: F401
from cadrays_amd import scenes
from cadrays_amd.view import View
from cadrays_amd._lib import load_library
lib = load_library()
v = View(0).load_scene(scenes.baseline_config(a.config))
buf = (C.c_ulonglong * 16)()
for i in range(3):
    v.reset(); v.sync(); lib.crh_exp_frame_stats(buf)
    t = time.perf_counter(); v.Redraw(); v.sync(); ms = (time.perf_counter() - t) * 1e3
    lib.crh_exp_frame_stats(buf)
    s = list(buf); st = v.stats()
    turns, have, dry, inner, tri, don, calls, sh_b, sh_n, spin_f, spin_t = s[:11]
    print(f"frame {ms:.2f} ms  rays {st['rays_nearest'] + st['rays_any']}  engine calls {calls}  turns {turns}  rays held per turn {have / max(turns, 1):.1f}  dry turns {dry / max(turns, 1):.2f}  "
          f"donation blocks per turn {don / max(turns, 1):.2f}  inner steps {inner} ({inner / max(st['rays_nearest'] + st['rays_any'], 1):.1f} per ray, {inner / max(turns, 1):.1f} per turn of <= 128)  "
          f"triangle tests {tri} ({tri / max(turns, 1):.1f} per turn)  shade batches {sh_b} x {sh_n / max(sh_b, 1):.1f} paths  idle spins feeder {spin_f} tracers {spin_t}")
