#!/usr/bin/env python3
"""Hunt: tests/test_gpu_fuzz.py::test_random_walks_over_the_static_moved_split with many more seeds (static tree + moved objects, two-pass and one-walk
traversal, objects dragged away and put back; oracle in step after every call).   python tests/hunts/split_walks.py [first] [last]"""
import importlib.util, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch  # noqa: F401
spec = importlib.util.spec_from_file_location("fz", "tests/test_gpu_fuzz.py"); fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
from cadrays_amd.view import View
from oracle.pyoracle import Oracle
a, b = (int(sys.argv[1]) if len(sys.argv) > 1 else 16), (int(sys.argv[2]) if len(sys.argv) > 2 else 600)
bad = []
for seed in range(a, b):
    try:
        fz.test_random_walks_over_the_static_moved_split(View, Oracle, seed)
    except AssertionError as e:
        bad.append((seed, str(e)[:120]))
print(f"{b - a} split walks, mismatches:", bad)
