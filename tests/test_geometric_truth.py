"""Independent geometric ground truth for the traversal ("parity is partial"): the quantised-box BVH walk of
BOTH sides (CPU oracle here, gfx950 kernels under -m gpu) against exhaustive ray x triangle tests that share no code with either
(oracle/brute_force.c: no BVH, no crh_* header).

  * f32 leg: the product's triangle formula over ALL triangles.  The BVH walk must return bit-identical (t, u, v) on every ray,
    grazing ones included -- the boxes are evaluated as quantised planes in t-space with rounding, and this is the check that
    they never cull a triangle the ray hits ("zero misses").
  * f64 leg: Moeller-Trumbore in double precision, another formula.  Same triangle, or |dt| <= 1e-4 * max(1, t); disagreement
    is tolerated only where the double-precision hit lies within 1e-5 (barycentric) of a triangle edge, i.e. where float and
    double may legitimately fall on different sides of the edge -- and never on rays that were not aimed at an edge.
Rays: uniformly random, axis-parallel (with -0.0 components), and rays aimed at triangle edges / vertices +- 1e-6 of the triangle's
size.  Scene scales 0.01 ... 50; single-level and two-level (per-object transforms) trees.
"""
import ctypes as C
import dataclasses
import os
import subprocess

import numpy as np
import pytest

from cadrays_amd import scenes
from cadrays_amd.materials import BSDF

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(HERE), "oracle")
_BF = None


def bf():
    global _BF
    if _BF is None:
        so, src = os.path.join(ORACLE_DIR, "libbrute_force.so"), os.path.join(ORACLE_DIR, "brute_force.c")
        if not os.path.exists(so) or os.path.getmtime(src) > os.path.getmtime(so):
            subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "-B", "libbrute_force.so"])
        _BF = C.CDLL(so)
    return _BF


def brute_f64(tri_pos, rays):
    n = len(rays)
    tuv, idx, margin = np.empty((n, 3), np.float64), np.empty(n, np.int32), np.empty(n, np.float64)
    bf().bf_nearest_f64(tri_pos.ctypes.data_as(C.POINTER(C.c_float)), C.c_uint32(len(tri_pos)), rays.ctypes.data_as(C.POINTER(C.c_float)),
                        C.c_uint32(n), tuv.ctypes.data_as(C.POINTER(C.c_double)), idx.ctypes.data_as(C.POINTER(C.c_int32)),
                        margin.ctypes.data_as(C.POINTER(C.c_double)))
    return tuv, idx, margin


def brute_f32(tri_pos, rays):
    out = np.empty((len(rays), 4), np.float32)
    bf().bf_nearest_f32(tri_pos.ctypes.data_as(C.POINTER(C.c_float)), C.c_uint32(len(tri_pos)), rays.ctypes.data_as(C.POINTER(C.c_float)),
                        C.c_uint32(len(rays)), out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def soup(n_tris, seed, scale):
    pos, nrm, tri = scenes.gen_scene(n_tris, seed, 1)
    return (pos * np.float32(scale)).astype(np.float32), nrm, tri


def make_rays(pos, tri, n, seed, scale):
    """(rays[n, 8], aimed[n]): aimed marks the rays constructed to graze a triangle edge or vertex"""
    r = np.random.default_rng(seed)
    v = pos[tri[:, :3]].astype(np.float64)                     # (nT, 3, 3)
    n_rand, n_axis = n // 2, n // 8
    n_graze = n - n_rand - n_axis
    org = (r.random((n_rand, 3)) * 2.4 - 1.2) * scale
    d = r.normal(size=(n_rand, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = [np.concatenate([org, d], 1)]
    # axis-parallel, with exact zeros of both signs in the other two components
    ax = r.integers(0, 3, n_axis); sg = r.choice([-1.0, 1.0], n_axis)
    da = np.zeros((n_axis, 3)); da[np.arange(n_axis), ax] = sg
    zs = r.choice([0.0, -0.0], (n_axis, 3)); da = np.where(da == 0, zs, da)
    oa = (r.random((n_axis, 3)) * 2.0 - 1.0) * scale
    k = r.integers(0, len(v), n_axis)                          # through a triangle's interior, so that most of them hit something
    w = r.dirichlet((1, 1, 1), n_axis)
    through = np.einsum("ij,ijk->ik", w, v[k])
    oa = np.where(np.abs(da) == 1.0, oa - 2.5 * scale * da, through)
    rays.append(np.concatenate([oa, da], 1))
    # grazing: a point on an edge (or a vertex), moved by +-1e-6 of the triangle's size along the in-plane edge normal
    k = r.integers(0, len(v), n_graze)
    e = r.integers(0, 3, n_graze)
    a, b, c = v[k, e], v[k, (e + 1) % 3], v[k, (e + 2) % 3]
    s = r.random(n_graze); s[: n_graze // 4] = 0.0            # a quarter at the vertex itself
    p = a + (b - a) * s[:, None]
    nrm = np.cross(b - a, c - a)
    inpl = np.cross(nrm, b - a); inpl /= np.maximum(np.linalg.norm(inpl, axis=1, keepdims=True), 1e-300)   # points towards c (inside)
    size = np.linalg.norm(b - a, axis=1)
    off = r.choice([-1e-6, 0.0, 1e-6], n_graze) * size
    target = p + inpl * off[:, None]
    dg = r.normal(size=(n_graze, 3)); dg /= np.linalg.norm(dg, axis=1, keepdims=True)
    og = target - dg * (r.random(n_graze)[:, None] * 1.5 + 0.2) * scale
    rays.append(np.concatenate([og, dg], 1))
    od = np.concatenate(rays, 0)
    out = np.zeros((n, 8), np.float32)
    out[:, :3] = od[:, :3]; out[:, 3] = 1e30; out[:, 4:7] = od[:, 3:]
    aimed = np.zeros(n, bool); aimed[n_rand + n_axis:] = True
    return out, aimed


def _inside_distance(tri_pos, rays, rows, prims):
    """double precision: how far inside (+) / outside (-) triangle prims[i] does ray rows[i] cross the triangle's plane (a
    distance), and the width of the band around an edge inside which float and double may disagree"""
    scene = float(np.abs(tri_pos).max())
    tp = tri_pos[prims].astype(np.float64).reshape(-1, 3, 3)
    o, d = rays[rows, :3].astype(np.float64), rays[rows, 4:7].astype(np.float64)
    nrm = np.cross(tp[:, 1] - tp[:, 0], tp[:, 2] - tp[:, 0])
    nl = np.maximum(np.linalg.norm(nrm, axis=1), 1e-300)
    t = np.einsum("ij,ij->i", tp[:, 0] - o, nrm) / np.einsum("ij,ij->i", d, nrm)
    P = o + d * t[:, None]
    dist = np.full(len(rows), np.inf)
    for e in range(3):
        a, b = tp[:, e], tp[:, (e + 1) % 3]
        inpl = np.cross(nrm / nl[:, None], b - a)                       # in-plane normal of the edge, pointing inside
        inpl /= np.maximum(np.linalg.norm(inpl, axis=1, keepdims=True), 1e-300)
        dist = np.minimum(dist, np.einsum("ij,ij->i", P - a, inpl))
    size = np.max(np.linalg.norm(tp - np.roll(tp, 1, axis=1), axis=2), axis=1)
    cosi = np.abs(np.einsum("ij,ij->i", d, nrm)) / nl                   # grazing incidence stretches the band along the plane:
    return dist, (2e-5 * size + 2e-6 * scene + 1e-6 * np.abs(t)) / np.maximum(cosi, 1e-5)      # a ray nearly IN the plane is ill-conditioned for any float test


def check_against_truth(hits, tri_pos, rays, aimed, exact_f32):
    """hits: (n, 4) {t, u, v, caller's triangle index} from a BVH walk"""
    n = len(rays)
    prim = hits[:, 3].view(np.int32)
    # ---- float leg: identical arithmetic without a BVH.  The walk is conservative with respect to the EXACT ray (guard band of
    # the slab test); the float triangle test, however, also accepts rays that pass just outside a
    # triangle (its own rounding, amplified at grazing incidence).  So a triangle the exhaustive test reports and the walk does
    # not reach must be such a false positive: in double precision the ray passes OUTSIDE it.  Everything else is bit-identical.
    n_fp = 0
    if exact_f32:
        ref = brute_f32(tri_pos, rays)
        rp = ref[:, 3].view(np.int32)
        same = (hits[:, :3].view(np.uint32) == ref[:, :3].view(np.uint32)).all(1)
        rows = np.where(~same)[0]
        n_fp = len(rows)
        if n_fp:
            assert (rp[rows] >= 0).all(), "the walk reports a hit the exhaustive float test does not know"
            dist, band = _inside_distance(tri_pos, rays, rows, rp[rows])
            assert (dist < 0).all(), f"{(dist >= 0).sum()} rays TRULY hit a triangle that the BVH walk culled (first: ray {rows[np.argmax(dist >= 0)]})"
            assert (dist > -band).all(), "a triangle far from the ray was accepted by the exhaustive float test?"
            assert aimed[rows].all() and n_fp < 2e-3 * max(aimed.sum(), 1), f"{n_fp} false positives of the float triangle test were culled"
        differ = same & (prim != rp)                             # only exact-t ties may name another triangle
        assert (hits[differ, 0] == ref[differ, 0]).all() and differ.mean() < 1e-3
    # ---- double leg: geometric truth
    tuv, idx, margin = brute_f64(tri_pos, rays)
    t64 = tuv[:, 0]
    hit64, hit = idx >= 0, prim >= 0
    tol = 1e-4 * np.maximum(1.0, np.where(hit64, t64, 1.0))
    agree = np.where(hit64 & hit, (prim == idx) | (np.abs(hits[:, 0].astype(np.float64) - t64) <= tol), hit64 == hit)
    # A hit the walk found and the truth does not (or the other way round) is acceptable only AT AN EDGE: in double precision,
    # how far inside (+) or outside (-) its triangle does the ray cross the triangle's plane, as a distance?  Float coordinates
    # (ulp 6e-8 of the scene size), the ray's rounded direction over its length and the float test itself blur an edge by about
    # 1e-6 of the scene size; a band of 2e-5 x triangle size + 2e-6 x scene size is generous and still far below any real miss.
    bad = ~agree
    if bad.any():
        scene = float(np.abs(tri_pos).max())

        def inside_distance(rows, prims):
            return _inside_distance(tri_pos, rays, rows, prims)

        excusable = np.zeros(n, bool)
        rows = np.where(bad & hit)[0]                                          # the walk's triangle: (nearly) crossed in double too?
        if len(rows):
            dist, band = inside_distance(rows, prim[rows])
            excusable[rows] = dist > -band
        rows = np.where(bad & hit64 & ~excusable)[0]                           # the truth's triangle: crossed at its very edge?
        if len(rows):
            dist, band = inside_distance(rows, idx[rows])
            excusable[rows] = dist < band
        assert not (bad & ~excusable).any(), f"{(bad & ~excusable).sum()} rays disagree with the double-precision truth away from any edge (first: ray {np.argmax(bad & ~excusable)})"
        assert (bad & ~aimed).mean() < 1e-4, "random rays disagree with the truth"
    return n_fp, int(bad.sum())


CPU_CASES = [(2500, 11, 1.0), (2500, 12, 0.01), (2500, 13, 50.0)]


@pytest.mark.parametrize("n_tris,seed,scale", CPU_CASES)
def test_oracle_walk_equals_exhaustive_tests(oracle_lib, n_tris, seed, scale):
    pos, nrm, tri = soup(n_tris, seed, scale)
    o = oracle_lib.Oracle().load_scene(scenes.Scene(pos, nrm, tri, [BSDF.CreateDiffuse(0.5)]))
    rays, aimed = make_rays(pos, tri, 6000, seed, scale)
    check_against_truth(o.trace_nearest(rays), np.ascontiguousarray(pos[tri[:, :3]].reshape(-1, 9)), rays, aimed, True)


def grouped(pos, nrm, tri, n_obj, seed):
    """the soup cut into n_obj objects (by centroid cell), every object given a rigid / scaled transform; returns the two-level
    scene and the world-space triangle corners (float64 transform of the object-space vertices)"""
    r = np.random.default_rng(seed)
    cen = pos[tri[:, :3]].mean(1)
    g = int(round(n_obj ** (1 / 3)))
    lo, hi = cen.min(0), cen.max(0)
    cell = np.minimum(((cen - lo) / (hi - lo + 1e-30) * g).astype(int), g - 1)
    obj = (cell[:, 0] * g + cell[:, 1]) * g + cell[:, 2]
    ext = float(np.abs(pos).max())
    xf = np.zeros((g ** 3, 12), np.float32)
    world = np.empty((len(tri), 3, 3), np.float64)
    for ob in range(g ** 3):
        a = r.normal(size=3); a /= np.linalg.norm(a)
        ang = r.random() * 2 * np.pi if ob % 3 else 0.0          # every third object is only translated (the kernel's fast entry)
        cs, sn = np.cos(ang), np.sin(ang); x, y, z = a
        R = np.array([[cs + x * x * (1 - cs), x * y * (1 - cs) - z * sn, x * z * (1 - cs) + y * sn],
                      [y * x * (1 - cs) + z * sn, cs + y * y * (1 - cs), y * z * (1 - cs) - x * sn],
                      [z * x * (1 - cs) - y * sn, z * y * (1 - cs) + x * sn, cs + z * z * (1 - cs)]]) * (1.0 if ob % 3 == 0 else r.uniform(0.6, 1.4))
        M = np.concatenate([R, (r.normal(size=3) * 0.05 * ext)[:, None]], 1).astype(np.float32)
        xf[ob] = M.reshape(12)
        sel = obj == ob
        Md = M.astype(np.float64)
        world[sel] = pos[tri[sel, :3]].astype(np.float64) @ Md[:, :3].T + Md[:, 3]
    sc = scenes.Scene(pos, nrm, tri, [BSDF.CreateDiffuse(0.5)], tri_object=obj.astype(np.int32), obj_xform=xf)
    return sc, world


def test_oracle_two_level_walk_against_double_precision_truth(oracle_lib):
    pos, nrm, tri = soup(3000, 21, 1.0)
    sc, world = grouped(pos, nrm, tri, 27, 5)
    wpos = world.astype(np.float32)
    rays, aimed = make_rays(wpos.reshape(-1, 3), np.arange(3 * len(tri)).reshape(-1, 3), 6000, 22, 1.0)
    o = oracle_lib.Oracle().load_scene(sc)
    # transforms round differently from the host's double-precision flattening: compare with tolerance only
    check_against_truth(o.trace_nearest(rays), np.ascontiguousarray(wpos.reshape(-1, 9)), rays, aimed, False)


GPU_CASES = [(20000, 31, 1.0), (20000, 32, 0.01), (20000, 33, 50.0), (20000, 34, 7.5)]


@pytest.mark.gpu
@pytest.mark.parametrize("n_tris,seed,scale", GPU_CASES)
def test_gpu_walk_equals_exhaustive_tests_200k_rays(hip_lib, n_tris, seed, scale):
    from cadrays_amd.view import View
    pos, nrm, tri = soup(n_tris, seed, scale)
    v = View(0).load_scene(scenes.Scene(pos, nrm, tri, [BSDF.CreateDiffuse(0.5)]))
    rays, aimed = make_rays(pos, tri, 200000, seed, scale)
    check_against_truth(v.trace_nearest(rays), np.ascontiguousarray(pos[tri[:, :3]].reshape(-1, 9)), rays, aimed, True)
    # any-hit agrees with nearest-hit about visibility (tmax = infinity)
    vis = v.trace_any(rays[:50000])
    assert np.array_equal(vis == 0, v.trace_nearest(rays[:50000])[:, 3].view(np.int32) >= 0)


@pytest.mark.gpu
@pytest.mark.parametrize("n_obj,seed,scale", [(27, 41, 1.0), (512, 42, 0.05), (64, 43, 20.0)])
def test_gpu_two_level_walk_against_double_precision_truth(hip_lib, n_obj, seed, scale):
    from cadrays_amd.view import View
    pos, nrm, tri = soup(20000, seed, scale)
    sc, world = grouped(pos, nrm, tri, n_obj, seed)
    wpos = world.astype(np.float32)
    rays, aimed = make_rays(wpos.reshape(-1, 3), np.arange(3 * len(tri)).reshape(-1, 3), 200000, seed + 1, scale)
    v = View(0).load_scene(sc)
    check_against_truth(v.trace_nearest(rays), np.ascontiguousarray(wpos.reshape(-1, 9)), rays, aimed, False)
