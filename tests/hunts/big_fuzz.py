#!/usr/bin/env python3
"""One-off hunt: 400 more seeded random scenes (tests/test_gpu_fuzz.random_scene) through the HIP path and the oracle; prints the seeds
whose images or counters differ (none so far).  Run on a GPU box: python tests/hunts/big_fuzz.py"""
import sys, importlib.util, numpy as np
sys.path.insert(0, '.')
import torch
spec = importlib.util.spec_from_file_location("fz", "tests/test_gpu_fuzz.py"); fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
from cadrays_amd.view import View
from oracle.pyoracle import Oracle
bad = []
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 1000, int(sys.argv[2]) if len(sys.argv) > 2 else 1400):
    sc = fz.random_scene(seed)
    v = View(0).load_scene(sc); v.enable_counters(True); v.reset(); o = Oracle().load_scene(sc)
    v.render(2); o.render(2)
    ok = np.array_equal(fz.bits(v.read_hdr()), fz.bits(o.read_hdr())) and v.stats()["nodes_nearest"] == o.stats()["nodes_nearest"] and v.stats()["tris_any"] == o.stats()["tris_any"]
    w = View(0).load_scene(sc); w.render(1); w.render(1)              # counters off: the small-batch schedule (tile ranges on two streams, small grids)
    ok = ok and np.array_equal(fz.bits(w.read_hdr()), fz.bits(o.read_hdr()))
    if not ok: bad.append(seed)
    v.close(); w.close(); o.close()
print("scenes checked, mismatches:", bad)
