#!/usr/bin/env python3
"""What the frame kernel's engines did in one lone frame, from an instrumented build (tools/ab_build.sh fstats "-DCRH_FRAME_STATS=1"):
   CRH_LIB_PATH=cadrays_amd/variants/fstats.so python tools/frame_stats.py [--config C3]"""
import argparse, ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ap = argparse.ArgumentParser(); ap.add_argument("--config", default="C3")
a = ap.parse_args()
import torch  # noqa: F401
from cadrays_amd import scenes
from cadrays_amd.view import View
from cadrays_amd._lib import load_library
lib = load_library()
v = View(0).load_scene(scenes.baseline_config(a.config))
buf = (C.c_ulonglong * 32)()
for i in range(3):
    v.reset(); v.sync(); lib.crh_exp_frame_stats(buf)
    t = time.perf_counter(); v.Redraw(); v.sync(); ms = (time.perf_counter() - t) * 1e3
    lib.crh_exp_frame_stats(buf)
    s = list(buf); st = v.stats()
    turns, have, dry, inner, tri, don, calls, sh_b, sh_n, spin_f, spin_t = s[:11]
    print(f"frame {ms:.2f} ms  rays {st['rays_nearest'] + st['rays_any']}  engine calls {calls}  turns {turns}  rays held per turn {have / max(turns, 1):.1f}  dry turns {dry / max(turns, 1):.2f}  "
          f"donation blocks per turn {don / max(turns, 1):.2f}  inner steps {inner} ({inner / max(st['rays_nearest'] + st['rays_any'], 1):.1f} per ray, {inner / max(turns, 1):.1f} per turn of <= 128)  "
          f"triangle tests {tri} ({tri / max(turns, 1):.1f} per turn)  shade batches {sh_b} x {sh_n / max(sh_b, 1):.1f} paths  idle spins feeder {spin_f} tracers {spin_t}")
    # round 6 (round-5 verdict, item 5): lanes per execution of each step, and the tracers' wave cycles by phase
    inner_w, tri_w, store_w, fin, big = s[11], s[12], s[13], s[14], s[15]
    c_refill, c_don, c_inner, c_leaf, c_retire, c_shade, c_gen = s[16], s[17], s[18], s[19], s[20], s[21], s[22]
    c_tr, c_fd = s[24], s[25]
    tr_busy = c_refill + c_don + c_inner + c_leaf + c_retire
    print(f"   lanes per execution: inner step {inner / max(inner_w, 1):.1f} of 64 ({inner_w} executions, {inner_w / max(turns, 1):.2f} per turn)   triangle step {tri / max(tri_w, 1):.1f} ({tri_w}, {tri_w / max(turns, 1):.2f} per turn)   "
          f"retire / store {fin / max(store_w, 1):.1f} ({store_w} turns of {turns} retire something)   shading {sh_n / max(sh_b, 1):.1f} per batch, {big / max(sh_b, 1):.1f} of them in the batch's largest material class")
    print(f"   tracer wavefronts: {c_tr / 1e6:.1f} M wave cycles in the kernel, {tr_busy / 1e6:.1f} M inside the engine = refill / claim {c_refill / max(tr_busy, 1):.3f}  donation {c_don / max(tr_busy, 1):.3f}  inner steps {c_inner / max(tr_busy, 1):.3f}  "
          f"leaf {c_leaf / max(tr_busy, 1):.3f}  retire / store {c_retire / max(tr_busy, 1):.3f};   feeders: {c_fd / 1e6:.1f} M wave cycles, shading {c_shade / max(c_fd, 1):.3f} generating {c_gen / max(c_fd, 1):.3f} of them (tracers that shade or generate are in these two as well)")
    # round 6: the frame's time line (100 MHz wall clock): start .. the slot cursor runs out (first / last workgroup to notice) .. first / last wavefront leaves
    M = (1 << 64) - 1
    if s[26] and s[29]:
        t0 = M - s[26]; out_first = M - s[27]; out_last = s[28]; end_last = s[29]; end_first = M - s[30]
        us = lambda t: (t - t0) / 100.0
        print(f"   time line (us after the first wavefront's start): slot cursor exhausted {us(out_first):.0f} (first workgroup to see it) .. {us(out_last):.0f} (last);  wavefronts leave {us(end_first):.0f} .. {us(end_last):.0f}"
              f"   => no new path after {us(out_first) / max(us(end_last), 1e-9):.2f} of the frame; the drain is {us(end_last) - us(out_first):.0f} us")

