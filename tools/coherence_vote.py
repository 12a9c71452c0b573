#!/usr/bin/env python3
"""Would the rays of bounce >= 1 form packets?  (round-5 verdict, item 2: measured, not argued.)  From an instrumented build:

   tools/ab_build.sh coh "-DCRH_COHERENCE_STATS=1"
   CRH_LIB_PATH=cadrays_amd/variants/coh.so python tools/coherence_vote.py [C3 C2 C1 CAD1M]

k_shade writes the survivors and the shadow rays of a chunk in rank order, so 64 consecutive ranks are the wavefront that traces them at the next bounce.
Per config (one wide step at the bench's samples per call) and per bounce: the fraction of such groups in which >= 48 of the 64 rays leave the SAME
triangle -- the precondition of a packet walk like k_trace_packets -- and how many of those leave through a delta lobe (the same direction up to the
sub-pixel jitter of the camera ray), for continuation rays and for shadow rays."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
from cadrays_amd import abi, scenes
from cadrays_amd.view import View
from cadrays_amd._lib import load_library
lib = load_library()
if not hasattr(lib, "crh_exp_coherence"):
    sys.exit("this library was not built with -DCRH_COHERENCE_STATS=1 (CRH_LIB_PATH=cadrays_amd/variants/coh.so)")
SPP = {"C3": 512, "C2": 256, "C1": 1024, "CAD1M": 512, "C5": 256}
buf = (C.c_ulonglong * 256)()
for cfg in (sys.argv[1:] or ["C3", "C2", "C1"]):
    sc = scenes.baseline_config(cfg)
    v = View(0).load_scene(sc); v.set_schedule(abi.SCHEDULE_WIDE)
    tiles = np.arange(v.n_tiles(), dtype=np.uint32)
    v.render_tiles(tiles, 0, SPP[cfg]); v.sync(); lib.crh_exp_coherence(buf)          # warm-up; counters cleared
    v.reset(); v.render_tiles(tiles, 0, SPP[cfg]); v.sync(); lib.crh_exp_coherence(buf)
    s = np.array(list(buf), np.float64).reshape(32, 8)
    print(f"== {cfg}: {len(sc.tri)} triangles, {sc.params.width}x{sc.params.height}, {SPP[cfg]} samples per pixel in one call, depth {sc.params.max_depth}")
    print("   rays INTO bounce | continuation rays | groups of 64 | same triangle >= 48 | ... through a delta lobe | shadow rays | groups | same triangle >= 48")
    tot = s.sum(0)
    for b in range(32):
        if s[b, 3] + s[b, 6] == 0:
            continue
        g, c, d, n, gs, cs, ns = s[b, 0], s[b, 1], s[b, 2], s[b, 3], s[b, 4], s[b, 5], s[b, 6]
        print(f"   {b + 1:17d} | {int(n):17d} | {int(g):12d} | {c / max(g, 1):19.4f} | {d / max(g, 1):23.4f} | {int(ns):11d} | {int(gs):6d} | {cs / max(gs, 1):.4f}")
    print(f"   all bounces >= 1: continuation groups coherent {tot[1] / max(tot[0], 1):.4f} (delta {tot[2] / max(tot[0], 1):.4f}), shadow groups coherent {tot[5] / max(tot[4], 1):.4f}; "
          f"share of all rays of the step that sit in coherent groups: {64 * (tot[1] + tot[5]) / max(tot[3] + tot[6] + sc.params.width * sc.params.height * SPP[cfg], 1):.4f}")
    v.close()
