#!/usr/bin/env python3
"""One-off hunt beside tests/hunts/big_fuzz.py: the random CALL-SEQUENCE tests of tests/test_gpu_fuzz.py with many more seeds (renders,
Redraws under look-ahead, tile subsets, resets, camera / material / light / environment / parameter changes, object moves,
adaptive on / off, checkpoints), GPU vs oracle after every step.  python tests/hunts/big_seq_fuzz.py [first] [last]"""
import os, sys, importlib.util
sys.path.insert(0, '.')
import torch  # noqa: F401
spec = importlib.util.spec_from_file_location("fz", "tests/test_gpu_fuzz.py"); fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
from cadrays_amd.view import View
from oracle.pyoracle import Oracle
a, b = (int(sys.argv[1]) if len(sys.argv) > 1 else 1000), (int(sys.argv[2]) if len(sys.argv) > 2 else 1200)
bad = []
for seed in range(a, b):
    if os.environ.get('CRH_FUZZ_VERBOSE'): print('seed', seed, file=sys.stderr, flush=True)
    for fn in (fz.test_random_call_sequences_keep_both_sides_in_step, fz.test_random_sequences_two_level_adaptive_checkpoint):
        try:
            fn.__wrapped__(View, Oracle, seed) if hasattr(fn, "__wrapped__") else fn(View, Oracle, seed)
        except AssertionError as e:
            bad.append((fn.__name__, seed, str(e)[:80]))
print(f"{b - a} seeds x 2 sequence kinds, mismatches:", bad)
