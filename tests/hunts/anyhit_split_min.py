"""Minimal form of the hunt: one context, split placement, COUNT any-hit of 100 000 short rays against the oracle.  Prints the number of wrong answers.
   CRH_LIB_PATH=<library> python tests/hunts/anyhit_split_min.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np, torch  # noqa
from test_two_level import object_scene
from test_trace_instantiations import hunt_rays, placement
import cadrays_amd
from cadrays_amd._lib import load_library
from cadrays_amd.view import View
from oracle import pyoracle
sc = object_scene(None, 128, 96)
rays, short = hunt_rays()
out = []
for kind in ("split", "all_moved"):
    o = pyoracle.Oracle().load_scene(sc); o.set_transforms(placement(kind)); want = o.trace_any(short)
    v = View(0).load_scene(sc); v.set_transforms(placement(kind))
    v.enable_counters(True); c = int((v.trace_any(short) != want).sum())
    v.enable_counters(False); p = int((v.trace_any(short) != want).sum())
    out.append(f"{kind}: COUNT {c} plain {p}")
    v.close(); o.close()
print(os.path.basename(os.environ.get("CRH_LIB_PATH", "default build")), " | ".join(out), flush=True)
