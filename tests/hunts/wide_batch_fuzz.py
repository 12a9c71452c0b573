#!/usr/bin/env python3
"""Hunt for the slot layout of wide batches (k_raygen / k_accumulate: 64 / G pixels x G samples per wavefront, LDS-tiled accumulate):
random small scenes rendered with sample counts that exercise every G (1 .. 64, multiples and non-multiples of 8 and 64), in one call, in
two calls, under look-ahead and through crh_render_tiles with a tile subset and a sample offset; image and counters against the oracle.
Every second seed runs under CRH_SCHEDULE_WIDE with a random path budget, so that the wide kernels (camera-ray packets on packet nodes or, for scenes
of objects, on the 64-byte nodes) and the cut of a call into tile groups x sample batches (crh_schedule.cpp) see the same variety.
    python tests/hunts/wide_batch_fuzz.py [first] [last]"""
import dataclasses, importlib.util, sys
sys.path.insert(0, '.')
import numpy as np
import torch  # noqa: F401
spec = importlib.util.spec_from_file_location("fz", "tests/test_gpu_fuzz.py"); fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
from cadrays_amd.view import View
from oracle.pyoracle import Oracle

a, b = (int(sys.argv[1]) if len(sys.argv) > 1 else 0), (int(sys.argv[2]) if len(sys.argv) > 2 else 60)
bad = []
for seed in range(a, b):
    r = np.random.default_rng(seed)
    sc = fz.random_scene(seed + 900)
    sc = dataclasses.replace(sc, params=dataclasses.replace(sc.params, width=int(r.integers(9, 50)), height=int(r.integers(9, 34)), max_depth=min(sc.params.max_depth, 4),
                                                            tile_size=int(r.choice([8, 16, 32]))))
    ns = int(r.choice([16, 24, 32, 40, 48, 64, 72, 96, 128, 130, 192]))
    v = View(0).load_scene(sc); o = Oracle().load_scene(sc)
    wide = seed % 2 == 1
    if wide:
        from cadrays_amd import abi
        ns = int(r.choice([64, 72, 128, 130, 192, 256, 320, 400]))
        v.set_schedule(abi.SCHEDULE_WIDE)
        tpp = sc.params.tile_size ** 2
        v.set_path_budget(max(1024, int(r.choice([3, 8, 20, 64, 1000])) * tpp * int(r.choice([64, 100, 128, 256]))))
    mode = int(r.integers(0, 4))
    if mode == 0:
        v.enable_counters(not wide or seed % 4 == 1); v.reset(); v.render(ns); o.render(ns)
    elif mode == 1:
        k = int(r.integers(1, ns)); v.render(k); v.render(ns - k); o.render(ns)
    elif mode == 2:
        v.set_lookahead(int(r.choice([16, 32, 64]))); done = 0
        while done < ns:
            k = int(r.integers(1, 20)); k = min(k, ns - done); v.render(k); done += k
        o.render(ns)
    else:
        tiles = np.arange(v.n_tiles(), dtype=np.uint32); sel = tiles[r.random(len(tiles)) < 0.6]
        if not len(sel): sel = tiles[:1]
        first = int(r.integers(0, 50)); v.render_tiles(sel, first, ns); o.render_tiles(sel, first, ns)
    ok = np.array_equal(fz.bits(v.read_hdr()), fz.bits(o.read_hdr()))
    if mode == 0 and (not wide or seed % 4 == 1): ok = ok and all(v.stats()[k] == o.stats()[k] for k in ("rays_nearest", "nodes_nearest", "tris_nearest", "rays_any", "samples"))
    if not ok: bad.append((seed, ns, mode)); print("MISMATCH", seed, ns, mode, flush=True)
    v.close(); o.close()
print(f"{b - a} wide batches, mismatches:", bad)
