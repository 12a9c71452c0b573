#!/usr/bin/env python3
"""Verdict r4 item 6, first step (CPU only): what would packed-convert FP8 planes cost in node visits?

gfx950 converts two FP8 values per instruction (v_cvt_pk_f32_fp8 / v_cvt_scalef32_pk_f32_fp8); the 24 v_cvt_f32_ubyteN of a node visit (20 % of its VALU
instructions) would become 12 if the child planes were E4M3 numbers instead of bytes.  E4M3 has three mantissa bits, so the planes are stored
corner-relative -- lower planes as offsets from the node's minimum corner, upper planes as offsets from its maximum corner, in grid steps, rounded so that the
box only grows -- where small offsets are exact (integers up to 16) and large ones coarse (steps of 16 between 128 and 255 grid steps).
This script takes the PRODUCT's tree of a BASELINE scene (the host builder, no GPU), re-quantises every child box that way, and walks both trees with the spec's
ordered nearest-hit traversal for camera rays and for incoherent rays: node visits and triangle tests per ray, byte planes against FP8 planes.

  python tools/experiments/fp8_planes_visits.py [--tris 1000000] [--rays 1500]"""
import argparse, math, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cadrays_amd import scenes
from cadrays_amd.view import build_bvh_host

ap = argparse.ArgumentParser(); ap.add_argument("--tris", type=int, default=1_000_000); ap.add_argument("--rays", type=int, default=1500)
a = ap.parse_args()

def e4m3_floor(x):
    """largest E4M3 value <= x (x >= 0, <= 448): 3 mantissa bits, integers exact up to 16"""
    x = np.asarray(x, np.float64)
    out = np.zeros_like(x)
    pos = x >= 2.0 ** -9                                  # smallest subnormal
    e = np.floor(np.log2(np.maximum(x, 2.0 ** -9)))
    e = np.clip(e, -6, 8)
    step = 2.0 ** (e - 3)
    out[pos] = (np.floor(x / step) * step)[pos]
    return np.minimum(out, 448.0)

sc = scenes.baseline_config("C3", n_tris=a.tris)
t0 = time.time()
nodes, order = build_bvh_host(sc.pos, sc.tri)
nw = nodes.view(np.uint32)
print(f"tree of {len(sc.tri)} triangles: {len(nodes)} nodes ({time.time() - t0:.1f} s)")
tri_v = sc.pos[sc.tri[order][:, :3]].astype(np.float64)         # leaf-order triangles [n, 3, 3]

# ---- decode the byte planes of every node: lo / hi [node, child, axis] in world units, plus the FP8 variant
w3 = nw[:, 3]
kexp = np.stack([((w3 >> (8 * ax)) & 0xff).astype(np.int8).astype(np.int32) for ax in range(3)], axis=1)     # [n, 3] signed exponents
step = np.ldexp(1.0, kexp)                                                                                    # [n, 3]
org = nodes[:, :3].astype(np.float64)
ni = (w3 >> 24) & 7; nc = (w3 >> 28) & 7
qlo = np.stack([np.stack([(nw[:, 4 + ax] >> (8 * k)) & 0xff for ax in range(3)], axis=1) for k in range(4)], axis=1).astype(np.float64)   # [n, 4, 3]
qhi = np.stack([np.stack([(nw[:, 7 + ax] >> (8 * k)) & 0xff for ax in range(3)], axis=1) for k in range(4)], axis=1).astype(np.float64)
valid = np.arange(4)[None, :] < nc[:, None]
lo8 = org[:, None, :] + qlo * step[:, None, :]; hi8 = org[:, None, :] + qhi * step[:, None, :]
# node extent in grid steps: the largest upper plane of a valid child (the node's maximum corner on its own grid)
ext = np.where(valid[:, :, None], qhi, 0).max(axis=1)                                                         # [n, 3]
off_lo = e4m3_floor(qlo)                                                                                      # lower planes: offsets from the minimum corner, rounded down
off_hi = e4m3_floor(ext[:, None, :] - qhi)                                                                    # upper planes: offsets from the maximum corner, rounded down (box grows)
lof = org[:, None, :] + off_lo * step[:, None, :]; hif = org[:, None, :] + (ext[:, None, :] - off_hi) * step[:, None, :]
grow = ((hif - lof) / np.maximum(hi8 - lo8, 1e-30))[valid]
print(f"child box edge, FP8 / byte planes: mean {grow.mean():.4f}, 90th percentile {np.percentile(grow, 90):.4f}, max {grow.max():.3f}")
child_base = nw[:, 10]; leaf_base = nw[:, 11] & 0x0FFFFFFF

def walk(o, d, lo, hi):
    """the spec's ordered walk without the guard band (both variants alike): visits, tests, hit distance"""
    inv = 1.0 / np.where(np.abs(d) < 1e-15, 1e-15, d)
    best = math.inf; visits = tests = 0
    stack = [0]
    while stack:
        ref = stack.pop()
        if ref < 0:                                       # leaf: ~ref = leaf position
            t = tri_v[~ref]; tests += 1
            e0 = t[1] - t[0]; e1 = t[0] - t[2]; n = np.cross(e1, e0); den = n @ d
            if den != 0.0:
                to = t[0] - o; tt = (n @ to) / den; vc = np.cross(d, to); uu = (vc @ e1) / den; vv = (vc @ e0) / den
                if tt >= 0 and uu >= 0 and vv >= 0 and uu + vv <= 1 and tt < best: best = tt
            continue
        visits += 1
        k = int(nc[ref])
        t1 = (lo[ref, :k] - o) * inv; t2 = (hi[ref, :k] - o) * inv
        tmin = np.maximum(np.minimum(t1, t2).max(axis=1), 0.0); tmax = np.minimum(np.maximum(t1, t2).min(axis=1), best)
        hitk = np.nonzero(tmin <= tmax)[0]
        for s in sorted(hitk, key=lambda s: (tmin[s], s), reverse=True):       # far .. near onto the stack: the nearest is popped first
            stack.append(int(child_base[ref] + s) if s < ni[ref] else ~int(leaf_base[ref] + (s - ni[ref])))
    return visits, tests, best

r = np.random.default_rng(7)
def camera_rays(n):
    cam = sc.camera; eye = np.array(cam.eye, np.float64); fwd = np.array(cam.dir, np.float64); up = np.array(cam.up, np.float64)
    right = np.cross(fwd, up); th = math.tan(math.radians(cam.fovy_deg) / 2); asp = sc.params.width / sc.params.height
    u = r.uniform(-1, 1, n); v = r.uniform(-1, 1, n)
    d = fwd[None] + right[None] * (u * th * asp)[:, None] + up[None] * (v * th)[:, None]
    return np.repeat(eye[None], n, 0), d / np.linalg.norm(d, axis=1, keepdims=True)
def scattered_rays(n):
    o = r.uniform(-0.9, 0.9, (n, 3)); d = r.normal(size=(n, 3))
    return o, d / np.linalg.norm(d, axis=1, keepdims=True)
for name, (O, D) in (("camera rays", camera_rays(a.rays)), ("incoherent rays from inside the scene", scattered_rays(a.rays))):
    res = {}
    for tag, (lo, hi) in (("byte planes", (lo8, hi8)), ("FP8 planes ", (lof, hif))):
        t0 = time.time(); tot = np.zeros(2); hits = []
        for i in range(len(O)):
            vi, te, b = walk(O[i], D[i], lo, hi); tot += (vi, te); hits.append(b)
        res[tag] = (tot / len(O), np.array(hits))
        print(f"{name:40s} {tag}: {tot[0] / len(O):7.2f} node visits, {tot[1] / len(O):6.2f} triangle tests per ray   ({time.time() - t0:.0f} s)")
    (v8, h8), (vf, hf) = res["byte planes"], res["FP8 planes "]
    assert np.array_equal(h8, hf), "conservative boxes must not change a hit"
    print(f"{'':40s} visits x {vf[0] / v8[0]:.3f}, tests x {vf[1] / v8[1]:.3f}; per-visit VALU 118 -> 106 (12 of 24 converts gone): traversal VALU x {vf[0] / v8[0] * 106 / 118:.3f}")
