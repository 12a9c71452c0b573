#!/usr/bin/env python3
"""Hunt: random PLY files (ascii / binary little endian; float or double coordinates; properties in any order; with or without
normals and s,t / u,v / texture_u,texture_v; extra properties and elements; polygons of 3..6 corners; every list count / index type)
read by the Python reader and by the C++ reader (host/model_tcl.hpp) through the same one-mesh model.tcl; prints the seeds whose
scenes differ in a byte.  CPU only.      python tools/fuzz_ply_reader.py [first] [last]"""
import os, shutil, struct, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cadrays_amd import scene_io
from cadrays_amd.scene_tcl import read_scene

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
exe = os.path.join(ROOT, "cadrays_amd", "host", "model_tcl_dump")
a, b = (int(sys.argv[1]) if len(sys.argv) > 1 else 0), (int(sys.argv[2]) if len(sys.argv) > 2 else 300)
T = {"char": "b", "uchar": "B", "short": "h", "ushort": "H", "int": "i", "uint": "I", "float": "f", "double": "d"}
bad = []
for seed in range(a, b):
    r = np.random.default_rng(seed)
    nv = int(r.integers(3, 60)); nf = int(r.integers(1, 80))
    ascii_ = bool(r.integers(0, 2)); ctype = str(r.choice(["float", "double"]))
    props = [("x", ctype), ("y", ctype), ("z", ctype)]
    if r.random() < 0.7: props += [("nx", "float"), ("ny", "float"), ("nz", "float")]
    if r.random() < 0.6:
        u, v = [("s", "t"), ("u", "v"), ("texture_u", "texture_v")][int(r.integers(0, 3))]; props += [(u, "float"), (v, "float")]
    if r.random() < 0.4: props += [("red", "uchar"), ("green", "uchar"), ("blue", "uchar")]
    if r.random() < 0.3: props += [("quality", str(r.choice(["float", "double", "short", "int"])))]
    order = r.permutation(len(props)); props = [props[i] for i in order]
    vals = {}
    for n_, t_ in props:
        if t_ in ("float", "double"): vals[n_] = (r.normal(size=nv) * 2).astype(np.float32 if t_ == "float" else np.float64)
        elif t_ == "uchar": vals[n_] = r.integers(0, 256, nv)
        else: vals[n_] = r.integers(-100, 100, nv)
    for k in ("nx", "ny", "nz"):
        if k in vals and r.random() < 0.1: vals[k][:] = 0                                         # degenerate normals do occur
    cnt_t = str(r.choice(["uchar", "ushort", "uint"])); idx_t = str(r.choice(["int", "uint", "ushort", "short"]))
    fname = str(r.choice(["vertex_indices", "vertex_index"]))
    faces = [list(r.integers(0, nv, int(r.integers(3, 7)))) for _ in range(nf)]
    extra_first = r.random() < 0.2; extra_last = r.random() < 0.3
    face_extra = r.random() < 0.2                                                                  # a scalar property beside the list
    hdr = ["ply", "format %s 1.0" % ("ascii" if ascii_ else "binary_little_endian"), "comment fuzz %d" % seed]
    if extra_first: hdr += ["element thing 2", "property float a", "property uchar b"]
    hdr += ["element vertex %d" % nv] + ["property %s %s" % (t_, n_) for n_, t_ in props]
    hdr += ["element face %d" % nf, "property list %s %s %s" % (cnt_t, idx_t, fname)] + (["property uchar flags"] if face_extra else [])
    if extra_last: hdr += ["element edge 3", "property int v1", "property int v2"]
    hdr += ["end_header"]
    body = bytearray(); lines = []
    def put(t_, x):
        if ascii_: lines[-1].append(repr(float(x)) if t_ in ("float", "double") else str(int(x)))
        else: body.extend(struct.pack("<" + T[t_], x if t_ in ("float", "double") else int(x)))
    if extra_first:
        for i in range(2): lines.append([]); put("float", 0.5 * i); put("uchar", i)
    for i in range(nv):
        lines.append([])
        for n_, t_ in props: put(t_, vals[n_][i])
    for f in faces:
        lines.append([]); put(cnt_t, len(f))
        for j in f: put(idx_t, j)
        if face_extra: put("uchar", 7)
    if extra_last:
        for i in range(3): lines.append([]); put("int", i); put("int", i + 1)
    tmp = tempfile.mkdtemp()
    try:
        with open(os.path.join(tmp, "m.ply"), "wb") as f:
            f.write(("\n".join(hdr) + "\n").encode())
            f.write(("\n".join(" ".join(l) for l in lines) + "\n").encode() if ascii_ else bytes(body))
        model = os.path.join(tmp, "model.tcl")
        open(model, "w").write("variable Root [file dirname [file normalize [info script]]]\nrtmeshread $Root/m.ply Mesh -group \nvdisplay Mesh -noupdate\nvbsdf Mesh -Kd 0.5 0.5 0.5 -noupdate\n")
        try:
            py, bld = read_scene(model, 32, 24)
        except Exception as e:
            py = None; perr = repr(e)[:100]
        pa, pb = os.path.join(tmp, "py.crhscene"), os.path.join(tmp, "cpp.crhscene")
        p = subprocess.run([exe, model, pb, "32x24"], capture_output=True, text=True)
        if py is None or p.returncode:
            if not (py is None and p.returncode):                                            # both refusing the same file is agreement
                bad.append(seed); print(seed, "python:", "ok" if py is not None else perr, "| c++:", p.returncode, p.stderr.strip()[:120], flush=True)
            continue
        scene_io.save_scene(py, pa)
        A, B = open(pa, "rb").read(), open(pb, "rb").read()
        if A != B:
            bad.append(seed); print(seed, "ascii" if ascii_ else "binary", ctype, [n for n, _ in props], cnt_t, idx_t, len(A), len(B), flush=True)
    finally:
        if os.environ.get("KEEP_TMP"): print("TMP", tmp)
        else: shutil.rmtree(tmp, ignore_errors=True)
print(f"{b - a} PLY files, mismatches:", bad)
