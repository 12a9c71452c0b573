#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the CPU oracle (NOT from the reference: the reference has no
implementation of this path on disk and no golden vectors -- SURVEY.md section 8c).  Purpose: freeze the
oracle (any drift in the spec shows up as a diff here) and give the GPU tests a fixture that does not
need the oracle at run time.  Re-run only on a deliberate spec change:  python tests/golden/make_golden.py
"""
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from cadrays_amd import scenes  # noqa: E402
from cadrays_amd.materials import BSDF  # noqa: E402
from oracle.pyoracle import Oracle  # noqa: E402


def crc(a):
    return np.uint32(zlib.crc32(np.ascontiguousarray(a).tobytes()))


def golden_rays(n=4096, seed=5):
    r = np.random.default_rng(seed)
    o = r.random((n, 3)) * 2 - 1
    d = r.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.zeros((n, 8), np.float32)
    rays[:, 0:3] = o; rays[:, 3] = 1e15; rays[:, 4:7] = d
    return rays


def soup_scene(n=5000):
    pos, nrm, tri = scenes.gen_scene(n, 1, 2)
    return scenes.Scene(pos, nrm, tri, [BSDF.CreateDiffuse(0.8), BSDF.Glossy()])


def render_cases():
    c3 = scenes.baseline_config("C3", 64, 36, n_tris=4000); c3.env = scenes.procedural_sky(128, 64, 1)
    return {
        "cornell_c1_64_spp8": (scenes.cornell_box(False, 64, 64), 8),        # BASELINE config C1 at fixture scale
        "cornell_full_64_spp4": (scenes.cornell_box(True, 64, 64), 4),
        "materials_80x60_spp2": (scenes.materials_scene(80, 60, 16, 8), 2),
        "c2_small_64x36_spp2": (scenes.baseline_config("C2", 64, 36, n_tris=4000), 2),
        "c3_small_64x36_spp2": (c3, 2),
    }


def main():
    out = {}
    sc = soup_scene()
    o = Oracle().load_scene(sc)
    nodes, tris = o.get_bvh()
    rays = golden_rays()
    out["soup_bvh_nodes_crc"] = crc(nodes); out["soup_bvh_tris_crc"] = crc(tris); out["soup_n_nodes"] = np.uint32(len(nodes))
    o.reset()
    out["soup_hits"] = o.trace_nearest(rays)
    st = o.stats()
    out["soup_counters"] = np.array([st["nodes_nearest"], st["tris_nearest"]], np.uint64)
    rs = rays.copy(); rs[:, 3] = 0.3
    out["soup_vis"] = o.trace_any(rs).astype(np.uint8)
    np.savez_compressed(os.path.join(HERE, "trace_soup5k.npz"), **out)
    for name, (scn, spp) in render_cases().items():
        o = Oracle().load_scene(scn)
        o.render(spp)
        st = o.stats()
        np.savez_compressed(os.path.join(HERE, f"render_{name}.npz"), hdr=o.read_hdr(), ldr_crc=crc(o.read_ldr()),
                            counters=np.array([st[k] for k in ("rays_nearest", "rays_any", "nodes_nearest", "tris_nearest", "nodes_any", "tris_any", "shaded_hits", "samples")], np.uint64))
    # elementary math: bit patterns on a fixed grid
    from oracle import pyoracle
    x = np.linspace(0, 1, 4097, dtype=np.float32)[:-1]
    s, c = pyoracle.math_fn(0, x)
    m = {"sin2pi": s, "cos2pi": c, "exp": pyoracle.math_fn(1, (x * 170 - 85).astype(np.float32))[0],
         "log": pyoracle.math_fn(2, np.exp(x * 60 - 30).astype(np.float32))[0],
         "pow": pyoracle.math_fn(3, x, (x[::-1] * 2000).astype(np.float32))[0],
         "acos": pyoracle.math_fn(4, (x * 2 - 1).astype(np.float32))[0],
         "atan2": pyoracle.math_fn(5, (x * 2 - 1).astype(np.float32), (x[::-1] * 2 - 1).astype(np.float32))[0],
         "rng": pyoracle.rng_stream(12345, 678, 64),
         "frame_seeds": np.array([pyoracle.frame_seed(1, i) for i in range(16)], np.uint32)}
    np.savez_compressed(os.path.join(HERE, "math.npz"), **m)
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
