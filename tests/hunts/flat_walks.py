#!/usr/bin/env python3
"""Hunt: tests/test_gpu_fuzz.py::test_random_sequences_between_one_tree_and_object_trees with many more seeds (scenes whose objects all sit
at the identity <-> object trees + top level, oracle in step after every call).   python tests/hunts/flat_walks.py [first] [last]"""
import importlib.util, sys
sys.path.insert(0, '.')
import torch  # noqa: F401
spec = importlib.util.spec_from_file_location("fz", "tests/test_gpu_fuzz.py"); fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
from cadrays_amd.view import View
from oracle.pyoracle import Oracle
a, b = (int(sys.argv[1]) if len(sys.argv) > 1 else 12), (int(sys.argv[2]) if len(sys.argv) > 2 else 1000)
bad = []
for seed in range(a, b):
    try:
        fz.test_random_sequences_between_one_tree_and_object_trees(View, Oracle, seed)
    except AssertionError as e:
        bad.append((seed, str(e)[:80]))
print(f"{b - a} walks, mismatches:", bad)
