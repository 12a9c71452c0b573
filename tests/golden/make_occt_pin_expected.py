#!/usr/bin/env python3
"""Regenerates tools/occt_pin/expected/ (this backend's own output of the pinning kit: every switch setting x three scenes, 64 frames, 128 x 128) with
the CPU oracle -- whose images are, bit for bit, the HIP path's (tests/test_occt_pin_kit.py::test_committed_expected_images_are_reproduced checks
exactly that on the GPU), so a deliberate spec change can be committed from a machine without a GPU.  Round 6 ran it for crh_spec.h #15 (display
gamma 2): the .pfm files did not move, the .png files did, and a folder for the new switch appeared.   python tests/golden/make_occt_pin_expected.py
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tools", "occt_pin"))

import pin_kit  # noqa: E402
from oracle.pyoracle import Oracle  # noqa: E402


class OracleView(Oracle):
    def __init__(self, device=0):
        super().__init__()

    def Redraw(self):
        self.render(1)

    def sync(self):
        pass

    def set_lookahead(self, k):
        pass


if __name__ == "__main__":
    exp = os.path.join(ROOT, "tools", "occt_pin", "expected")
    meta = json.load(open(os.path.join(exp, "kit.json")))
    pin_kit.make(exp, meta["frames"], tuple(meta["size"]), None, view_cls=OracleView)
    for d, _, files in os.walk(exp):                      # the frame rate of a CPU run is not a fixture
        for f in files:
            if f.endswith(".txt"):
                open(os.path.join(d, f), "w").write("0")
