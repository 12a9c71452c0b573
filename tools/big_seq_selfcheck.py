#!/usr/bin/env python3
"""GPU-only variant of tests/hunts/big_seq_fuzz.py: the oracle's seat is taken by a second context of the product on the plain schedule
(CRH_PIPELINE=0, CRH_DONATE=0, CRH_LANES=1, CRH_FRAME_KERNEL=0: one stream, no frames in flight, no work donation, the staged launches), so the random call sequences of
tests/test_gpu_fuzz.py run several times faster, two contexts share the GPU, and a crash or mismatch can only come from the HIP
side.    python tools/big_seq_selfcheck.py [first] [last]"""
import importlib.util, os, sys
sys.path.insert(0, '.')
import torch  # noqa: F401
spec = importlib.util.spec_from_file_location("fz", "tests/test_gpu_fuzz.py"); fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
from cadrays_amd.view import View


class PlainView(View):
    def __init__(self):
        keep = {k: os.environ.get(k) for k in ("CRH_PIPELINE", "CRH_DONATE", "CRH_LANES", "CRH_FRAME_KERNEL")}
        os.environ.update(CRH_PIPELINE="0", CRH_DONATE="0", CRH_LANES="1", CRH_FRAME_KERNEL="0")
        try:
            super().__init__(0)
        finally:
            for k, v in keep.items():
                if v is None: os.environ.pop(k, None)
                else: os.environ[k] = v


a, b = (int(sys.argv[1]) if len(sys.argv) > 1 else 1000), (int(sys.argv[2]) if len(sys.argv) > 2 else 3000)
bad = []
for seed in range(a, b):
    if os.environ.get("CRH_FUZZ_VERBOSE"): print("seed", seed, file=sys.stderr, flush=True)
    for fn in (fz.test_random_call_sequences_keep_both_sides_in_step, fz.test_random_sequences_two_level_adaptive_checkpoint):
        try:
            fn(View, PlainView, seed)
        except AssertionError as e:
            bad.append((fn.__name__, seed, str(e)[:80]))
print(f"{b - a} seeds x 2 sequence kinds, product vs product on the plain schedule, mismatches:", bad)
