#!/usr/bin/env python3
"""Hunt: tests/test_visibility.py::test_hip_fuzz_visibility_transforms_adaptive_checkpoint with many more seeds and longer sequences (Display / Erase,
manipulator moves, objects added to the running scene, adaptive sampling on / off, checkpoints; oracle in step after every call).
    python tests/hunts/visibility_walks.py [first] [last]"""
import importlib.util, os, sys, types
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401
spec = importlib.util.spec_from_file_location("tv", os.path.join(ROOT, "tests/test_visibility.py")); tv = importlib.util.module_from_spec(spec); spec.loader.exec_module(tv)
from oracle import pyoracle
a, b = (int(sys.argv[1]) if len(sys.argv) > 1 else 5000), (int(sys.argv[2]) if len(sys.argv) > 2 else 5300)
bad = []
for seed in range(a, b):
    try:
        tv.visibility_sequence(pyoracle, seed, steps=16)
    except Exception as e:                                       # a mismatch (AssertionError) or a call the product refused and the oracle did not
        bad.append(seed); print("seed", seed, type(e).__name__, str(e)[:200], flush=True)
print(f"{b - a} Display / Erase / add / move sequences, mismatches: {bad}")
sys.exit(1 if bad else 0)
