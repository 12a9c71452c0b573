#!/usr/bin/env python3
"""Memory-safety hunt for the C++ file readers (host/model_tcl.hpp: model.tcl, PLY, PNG; host/jpeg_baseline.hpp), which parse files a
user hands them: valid files are truncated, bit-flipped, spliced and over-written with random bytes and fed to an ASan + UBSan build of
model_tcl_dump.  Any outcome but "clean exit 0" or "error message, exit 1" is a finding (sanitizer report, signal, hang).  CPU only.
    g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-sanitize-recover=undefined -o /tmp/model_tcl_dump_asan cadrays_amd/host/model_tcl_dump.cpp -lz
    python tools/fuzz_readers_malformed.py [/tmp/model_tcl_dump_asan] [rounds] [seed]"""
import os, shutil, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from PIL import Image

exe = sys.argv[1] if len(sys.argv) > 1 else "/tmp/model_tcl_dump_asan"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 300
r = np.random.default_rng(int(sys.argv[3]) if len(sys.argv) > 3 else 1)
tmp = tempfile.mkdtemp()
y, x = np.mgrid[0:37, 0:51]
img = np.clip(np.stack([x * 5, y * 6, (x + y) * 3], -1), 0, 255).astype(np.uint8)
seeds = {}
for name, kw in (("a.jpg", dict(quality=80)), ("b.jpg", dict(quality=60, subsampling=2, progressive=True)), ("c.jpg", dict(quality=90, subsampling=1, restart_marker_blocks=2))):
    Image.fromarray(img).save(os.path.join(tmp, name), **kw); seeds[name] = open(os.path.join(tmp, name), "rb").read()
Image.fromarray(img).save(os.path.join(tmp, "d.png")); seeds["d.png"] = open(os.path.join(tmp, "d.png"), "rb").read()
Image.fromarray(np.dstack([img, img[..., :1]])).save(os.path.join(tmp, "e.png")); seeds["e.png"] = open(os.path.join(tmp, "e.png"), "rb").read()
Image.fromarray(img[..., 0]).convert("P").save(os.path.join(tmp, "f.png")); seeds["f.png"] = open(os.path.join(tmp, "f.png"), "rb").read()
ply_hdr = b"ply\nformat binary_little_endian 1.0\nelement vertex 4\nproperty float x\nproperty float y\nproperty float z\nproperty float nx\nproperty float ny\nproperty float nz\nproperty float s\nproperty float t\nelement face 2\nproperty list uchar int vertex_indices\nend_header\n"
ply = ply_hdr + np.arange(32, dtype=np.float32).tobytes() + bytes([3]) + np.array([0, 1, 2], np.int32).tobytes() + bytes([3]) + np.array([0, 2, 3], np.int32).tobytes()
seeds["g.ply"] = ply
seeds["h.ply"] = b"ply\nformat ascii 1.0\nelement vertex 3\nproperty float x\nproperty float y\nproperty float z\nelement face 1\nproperty list uchar int vertex_indices\nend_header\n0 0 0\n1 0 0\n0 1 0\n3 0 1 2\n"
tcl = ("variable Root [file dirname [file normalize [info script]]]\nrtmeshread $Root/m.ply Mesh -group \nvdisplay Mesh -noupdate\nvbsdf Mesh -Kd 0.5 0.25 0.125 -noupdate\n"
       "vbsdf Mesh -baseFresnel Conductor 1.5 2.5 -noupdate\nrttexture Mesh \"$Root/t.png\"\nrttexture Mesh -scale 2.0 3.0\nvlocation Mesh -rotation 0 0 0.7071 0.7071\nvlocation Mesh -location 1 2 3\n"
       "vcamera -perspective -fovy 40\nvviewparams -proj 0 -1 0\nvviewparams -up 0 0 1\nvviewparams -at 0 0 0\nvviewparams -eye 0 -5 0\nvlight clear\nvlight add positional position 1 2 3 smoothness 0.5 intensity 10\n"
       "rtlight 0 -color 1 0.5 0.25\nvtextureenv on $Root/e.jpg\nvrenderparams -ray -gi -rayDepth 7\n").encode()
findings = 0


def mutate(d):
    d = bytearray(d); k = r.integers(0, 6)
    if k == 0 and len(d) > 4: d = d[:int(r.integers(0, len(d)))]
    elif k == 1:
        for _ in range(int(r.integers(1, 8))): d[int(r.integers(0, len(d)))] ^= 1 << int(r.integers(0, 8))
    elif k == 2:
        i = int(r.integers(0, len(d))); n = int(r.integers(1, 16)); d[i:i + n] = bytes(r.integers(0, 256, n).astype(np.uint8))
    elif k == 3:
        i, j = sorted(int(v) for v in r.integers(0, len(d), 2)); d = d[:i] + d[j:]
    elif k == 4:
        i = int(r.integers(0, len(d))); d[i:i] = bytes(r.integers(0, 256, int(r.integers(1, 64))).astype(np.uint8))
    else:
        i = int(r.integers(0, max(1, len(d) - 4))); d[i:i + 4] = bytes([255, 255, 255, 255]) if r.random() < 0.5 else bytes(4)
    return bytes(d)


def run(args, what, src=None):
    global findings
    try:
        p = subprocess.run([exe] + args, capture_output=True, text=True, timeout=20, errors="replace")
    except subprocess.TimeoutExpired:
        findings += 1; print("HANG", what, flush=True); return
    if p.returncode not in (0, 1) or "Sanitizer" in p.stderr or "runtime error" in p.stderr:
        findings += 1; print("FINDING", what, p.returncode, p.stderr.strip()[-600:], flush=True)
        keep = src or args[1]
        shutil.copyfile(keep, os.path.join("/tmp", "finding_%d_%s" % (findings, os.path.basename(keep))))


# ---- deterministic leg: headers cut short INSIDE a segment.  The random mutations above start from full-size files and almost
# never produce a segment whose declared length is smaller than the fields the reader takes from it.
def image_case(data, what):
    path = os.path.join(tmp, "hdr_case.bin"); open(path, "wb").write(data)
    run(["--image", path, os.path.join(tmp, "o.raw")], what)


for tiny in ("FFD8FFDB000300", "FFD8FFC4000300", "FFD8FFC0000308", "FFD8FFDD0002", "FFD8FFDA0002", "FFD8FFDB0000", "FFD8FFC00001", "FFD8FFEE000341",
             "89504E470D0A1A0A0000000049484452", "89504E470D0A1A0A00000000494844520000000000000000", "89504E470D0A1A0A0000000549484452000000010000000000"):
    image_case(bytes.fromhex(tiny), "tiny " + tiny)
for name, data in seeds.items():
    if name.endswith(".ply"):
        continue
    for cut in range(2, min(len(data), 700)):              # every truncation point inside the headers
        image_case(data[:cut], f"cut {name}@{cut}")
    if name.endswith(".jpg"):                              # every segment with its declared length shrunk to 0 .. 8 (file kept whole)
        o = 2
        while o + 4 <= len(data) and data[o] == 0xFF and data[o + 1] not in (0xDA, 0xD9):
            L = (data[o + 2] << 8) | data[o + 3]
            for short in range(0, 9):
                image_case(data[:o + 2] + bytes([0, short]) + data[o + 4:], f"short {name}@{o}:{short}")
            o += 2 + L
    else:                                                  # PNG: every chunk with its length field shrunk to 0 .. 12
        o = 8
        while o + 12 <= len(data):
            L = int.from_bytes(data[o:o + 4], "big")
            for short in range(0, 13):
                image_case(data[:o] + short.to_bytes(4, "big") + data[o + 4:], f"short {name}@{o}:{short}")
            o += 12 + L

for it in range(rounds):
    for name, data in seeds.items():
        path = os.path.join(tmp, "m_" + name); open(path, "wb").write(mutate(data))
        if name.endswith(".ply"):
            d2 = os.path.join(tmp, "s"); os.makedirs(d2, exist_ok=True)
            shutil.copyfile(path, os.path.join(d2, "m.ply")); Image.fromarray(img).save(os.path.join(d2, "t.png")); Image.fromarray(img).save(os.path.join(d2, "e.jpg"))
            open(os.path.join(d2, "model.tcl"), "wb").write(tcl)
            run([os.path.join(d2, "model.tcl"), os.path.join(tmp, "o.crhscene"), "16x12"], "ply " + name, path)
        else:
            run(["--image", path, os.path.join(tmp, "o.raw")], "image " + name)
    d2 = os.path.join(tmp, "s"); os.makedirs(d2, exist_ok=True)
    open(os.path.join(d2, "m.ply"), "wb").write(seeds["g.ply"]); Image.fromarray(img).save(os.path.join(d2, "t.png")); Image.fromarray(img).save(os.path.join(d2, "e.jpg"))
    open(os.path.join(d2, "model.tcl"), "wb").write(mutate(tcl))
    run([os.path.join(d2, "model.tcl"), os.path.join(tmp, "o.crhscene"), "16x12"], "tcl", os.path.join(d2, "model.tcl"))
shutil.rmtree(tmp, ignore_errors=True)
print(f"{rounds} rounds x {len(seeds) + 1} inputs, findings: {findings}")
sys.exit(1 if findings else 0)
