#!/usr/bin/env python3
"""Hunt beside tests/test_gpu_parity.py::test_random_call_sequences_with_frames_in_flight (default schedule -- frames in flight, work
donation, two tile ranges -- against the plain one-stream schedule, GPU only): many more seeds.   python tools/big_inflight_fuzz.py [first] [last]"""
import importlib.util, sys
sys.path.insert(0, '.')
import torch  # noqa: F401
import pytest
spec = importlib.util.spec_from_file_location("tp", "tests/test_gpu_parity.py"); tp = importlib.util.module_from_spec(spec); spec.loader.exec_module(tp)
from cadrays_amd.view import View
a, b = (int(sys.argv[1]) if len(sys.argv) > 1 else 5), (int(sys.argv[2]) if len(sys.argv) > 2 else 205)
bad = []
for seed in range(a, b):
    mp = pytest.MonkeyPatch()
    try:
        tp.test_random_call_sequences_with_frames_in_flight(View, mp, seed)
    except AssertionError as e:
        bad.append((seed, str(e)[:80]))
    finally:
        mp.undo()
print(f"{b - a} sequences, mismatches:", bad)
