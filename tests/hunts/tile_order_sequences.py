#!/usr/bin/env python3
"""Hunt (round 6): the frame kernel's per-scene measurements -- feeder count, tile order -- under random call sequences at a size where they are active (>= 2^20 paths
per frame).  Two contexts on the same scene: one as shipped (CRH_TILE_ORDER on, feeders measured), one with CRH_TILE_ORDER=0 and CRH_FRAME_FEED=3; random bursts of
Redraw()s with and without waiting, restarts, camera moves, asynchronous read-backs, Display / Erase and moves of objects.  Every comparison is bit-exact (no pixel
depends on either measurement), and the final state is compared with the oracle on sampled tiles.
    python tests/hunts/tile_order_sequences.py [first] [last]"""
import dataclasses, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np
import torch  # noqa: F401
from cadrays_amd import scenes
from cadrays_amd.view import View
from oracle import pyoracle

a, b = (int(sys.argv[1]) if len(sys.argv) > 1 else 0), (int(sys.argv[2]) if len(sys.argv) > 2 else 40)


def bits(x): return np.ascontiguousarray(x, np.float32).view(np.uint32)


pos, nrm, tri, ob = scenes.gen_cad_like(30_000, 7, with_objects=True)
base = scenes.baseline_config("CAD1M", 1216, 896, n_tris=30_000)
base.env = scenes.procedural_sky(256, 128, 1)
nO = int(ob.max()) + 1
I12 = np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], np.float32)
sc = dataclasses.replace(base, pos=pos, nrm=nrm, tri=tri, tri_object=ob, obj_xform=np.tile(I12, (nO, 1)))
bad = []
for seed in range(a, b):
    r = np.random.default_rng(9000 + seed)
    # (reproduction aids: HUNT_LOG=1 prints every step before it runs; HUNT_V="CRH_TILE_ORDER=0,CRH_FRAME_FEED=3" gives the first context that environment too)
    LOG = os.environ.get("HUNT_LOG") == "1"
    for k in ("CRH_TILE_ORDER", "CRH_FRAME_FEED"): os.environ.pop(k, None)
    for kv in filter(None, os.environ.get("HUNT_V", "").split(",")): os.environ[kv.split("=")[0]] = kv.split("=")[1]
    v = View(0).load_scene(sc)
    os.environ["CRH_TILE_ORDER"] = "0"; os.environ["CRH_FRAME_FEED"] = "3"
    w = View(0).load_scene(sc)
    for k in ("CRH_TILE_ORDER", "CRH_FRAME_FEED"): os.environ.pop(k, None)
    if os.environ.get("HUNT_ONLY") in ("v", "w"):            # (reproduction aid: the other context is closed and replaced by the one that stays)
        if os.environ["HUNT_ONLY"] == "v": w.close(); w = v
        else: v.close(); v = w
    o = None
    cam = sc.camera; xf = np.tile(I12, (nO, 1)); vis = np.ones(nO, np.uint8)
    try:
        inflight = 0; adaptive_on = False
        for step in range(40):
            k = int(r.integers(0, 10 + (4 if os.environ.get("HUNT_MORE_OPS") else 0)))      # HUNT_MORE_OPS=1: + asynchronous read-backs, look-ahead, adaptive sampling, tile subsets
            if LOG: print("seed", seed, "step", step, "op", k, file=sys.stderr, flush=True)
            if k <= 3:                                              # lone frames: restart, one frame, wait
                for _ in range(int(r.integers(1, 12))):
                    for x in (v, w): x.reset(); x.Redraw(); x.sync()
            elif k == 4:                                            # a burst without waiting
                n = int(r.integers(1, 9))
                if LOG: print("   burst of", n, file=sys.stderr, flush=True)
                for x in (v, w):
                    for _ in range(n): x.Redraw()
                    if os.environ.get("HUNT_SYNC_AFTER_BURST"): x.sync(); print("   burst done on", "vw"[x is w], file=sys.stderr, flush=True)
            elif k == 5:                                            # a drag: restart every frame, frames in flight
                for i in range(int(r.integers(2, 10))):
                    cam = dataclasses.replace(cam, eye=tuple(np.float32(cam.eye) + np.float32(r.normal(size=3) * 0.01)))
                    for x in (v, w): x.set_camera(cam); x.reset(); x.Redraw()
            elif k == 6:
                vis = (r.random(nO) > 0.15).astype(np.uint8)
                for x in (v, w): x.set_visibility(vis)
            elif k == 7:
                xf = np.tile(I12, (nO, 1)); m = int(r.integers(0, nO)); xf[m, 3::4] += (0.05 * r.normal(size=3)).astype(np.float32)
                for x in (v, w): x.set_transforms(xf)
            elif k == 8:                                            # a wide call in between
                n = int(r.choice([2, 5, 16]))
                if LOG: print("   render", n, file=sys.stderr, flush=True)
                for x in (v, w): x.render(n)
            elif k == 10:                                           # every frame displayed: asynchronous read-backs two frames behind, compared as they arrive
                n = int(r.integers(3, 10)); got = {id(v): [], id(w): []}
                for x in (v, w):
                    for i in range(n):
                        x.Redraw()
                        if i >= 2: got[id(x)].append(x.read_ldr_end())
                        x.read_ldr_begin()
                    got[id(x)].append(x.read_ldr_end()); got[id(x)].append(x.read_ldr_end())
                assert all(np.array_equal(p, q) for p, q in zip(got[id(v)], got[id(w)])), f"displayed frames differ at step {step}"
            elif k == 11:
                la = int(r.choice([1, 1, 4, 16]))
                for x in (v, w): x.set_lookahead(la)
            elif k == 12:
                on = bool(r.integers(0, 2)); per = int(r.choice([64, 128, 512]))
                for x in (v, w): x.set_adaptive(on, per)
                adaptive_on = on
            elif k == 13:                                           # a tile subset with its own sample range (another batch size on the context's stream)
                if not adaptive_on:
                    sub = r.choice(v.n_tiles(), int(r.integers(1, 400)), replace=False).astype(np.uint32); ns_ = int(r.integers(1, 4))
                    for x in (v, w): x.render_tiles(sub, 0, ns_)
            else:
                assert np.array_equal(v.read_ldr(), w.read_ldr()), f"LDR differs at step {step}"
            if os.environ.get("HUNT_SYNC_EVERY"):
                for x in (v, w): x.sync(); print("   step", step, "done on", "vw"[x is w], file=sys.stderr, flush=True)
            if r.random() < 0.4:
                assert np.array_equal(bits(v.read_hdr()), bits(w.read_hdr())), f"HDR differs at step {step} (op {k})"
        gv, gw = v.stats(), w.stats()
        for key in ("rays_nearest", "rays_any", "shaded_hits", "samples"):
            assert gv[key] == gw[key], key
        # the oracle on the final state: sampled tiles of a fresh accumulation
        for x in (v, w): x.set_adaptive(False, 128); x.set_lookahead(1); x.reset(); x.Redraw(); x.Redraw()
        o = pyoracle.Oracle().load_scene(sc); o.set_camera(cam); o.set_transforms(xf); o.set_visibility(vis)
        sample = np.unique(np.linspace(0, o.n_tiles() - 1, 9).astype(np.uint32))
        o.render_tiles(sample, 0, 2)
        acc = o.read_accum(); mask = acc[..., 3] == 2
        g = v.read_hdr()
        assert mask.sum() > 0 and np.array_equal(bits(g[mask]), bits(acc[..., :3][mask])), "differs from the oracle"
        v.tile_order()
        print("seed", seed, "ok; tuning", v.frame_tuning()["feeders"], "tile order", v.tile_order_calls, flush=True)
    except AssertionError as e:
        bad.append(seed); print("seed", seed, "MISMATCH:", str(e)[:200], flush=True)
    finally:
        v.close()
        if w is not v: w.close()
        if o is not None: o.close()
print(f"{b - a} sequences with the frame kernel's measurements on against off, mismatches: {bad}")
sys.exit(1 if bad else 0)
