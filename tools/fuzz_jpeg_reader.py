#!/usr/bin/env python3
"""Hunt beside tests/test_scene_tcl.py::test_cpp_jpeg_reader_matches_pillow: random small JPEG files (sizes 1..90, random / smooth / flat
content, quality 1..100, three samplings, sequential / progressive, optimised tables, restart intervals, grey) through the C++
reader (cadrays_amd/host/jpeg_baseline.hpp) and through Pillow; prints every file whose pixels differ.  CPU only.
    python tools/fuzz_jpeg_reader.py [n] [seed]"""
import os, struct, subprocess, sys, tempfile
import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
exe = os.path.join(ROOT, "cadrays_amd", "host", "model_tcl_dump")
n, seed = (int(sys.argv[1]) if len(sys.argv) > 1 else 600), (int(sys.argv[2]) if len(sys.argv) > 2 else 77)
tmp = tempfile.mkdtemp()
rng = np.random.default_rng(seed); bad = 0
for it in range(n):
    w, h = int(rng.integers(1, 90)), int(rng.integers(1, 90))
    if rng.integers(0, 4) == 0: w = int(rng.integers(1, 6))
    kind = rng.integers(0, 3)
    if kind == 0: img = rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
    elif kind == 1:
        y, x = np.mgrid[0:h, 0:w]; img = np.clip(np.stack([x * 255 / max(w - 1, 1), y * 255 / max(h - 1, 1), (x + y) * 3 % 256], -1), 0, 255).astype(np.uint8)
    else: img = np.full((h, w, 3), int(rng.integers(0, 256)), np.uint8)
    kw = dict(quality=int(rng.integers(1, 101)), subsampling=int(rng.integers(0, 3)), progressive=bool(rng.integers(0, 2)), optimize=bool(rng.integers(0, 2)))
    if rng.integers(0, 3) == 0: kw["restart_marker_blocks"] = int(rng.integers(1, 9))
    grey = rng.integers(0, 5) == 0
    if grey: kw.pop("subsampling")
    p = os.path.join(tmp, "f.jpg"); Image.fromarray(img[..., 0] if grey else img).save(p, **kw)
    out = os.path.join(tmp, "o.raw")
    r = subprocess.run([exe, "--image", p, out], capture_output=True, text=True)
    want = np.asarray(Image.open(p).convert("RGB"))
    if r.returncode: bad += 1; print(it, w, h, kw, r.stderr.strip()); continue
    d = open(out, "rb").read(); W, H, ch = struct.unpack("<3I", d[:12]); got = np.frombuffer(d[12:], np.uint8).reshape(H, W, ch)
    if got.shape != want.shape or not np.array_equal(got, want): bad += 1; print(it, w, h, kw, "max diff", int(np.abs(got.astype(int) - want).max()))
print(f"{n} files, mismatches: {bad}")
