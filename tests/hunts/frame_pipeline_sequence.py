import os, sys, dataclasses, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch
from cadrays_amd import scenes
from cadrays_amd.view import View
def bits(a): return np.ascontiguousarray(a, np.float32).view(np.uint32)
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 4
sc = scenes.cornell_box(True, 1216, 896)
r0 = np.random.default_rng(seed)
ops = []
for _ in range(14):
    k = r0.integers(0, 9); ops.append((int(k), int(r0.integers(1, 6)), float(r0.random())))
def run(v, sync_every=False):
    outs = []
    for k, n, x in ops:
        if k <= 2:
            for _ in range(n): v.Redraw()
        elif k == 3:
            v.set_camera(dataclasses.replace(sc.camera, eye=(0.3 + 0.3 * x, -1.5, 0.5))); v.reset()
        elif k == 4:
            mats = [dataclasses.replace(m) for m in sc.materials]; mats[0] = dataclasses.replace(mats[0], Kd=np.float32([x, 0.5, 0.3])); v.set_materials(mats)
        elif k == 5: outs.append(v.read_hdr().copy())
        elif k == 6: v.render_tiles(np.arange(0, v.n_tiles(), 3, dtype=np.uint32), 40 + n, n)
        elif k == 7: v.set_lookahead(1 + (n % 3) * 3)
        else:
            v.set_adaptive(n % 2 == 0, 64 + 16 * n)
            for _ in range(2): v.Redraw()
            v.set_adaptive(False, 64)
        if sync_every: v.sync()
    for _ in range(3): v.Redraw()
    outs.append(v.read_hdr().copy())
    return outs
def ctx(**env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env); v = View(0).load_scene(sc)
    for k, val in old.items():
        if val is None: os.environ.pop(k, None)
        else: os.environ[k] = val
    return v
ref = run(ctx(CRH_PIPELINE="0", CRH_DONATE="0", CRH_LANES="1", CRH_FRAME_KERNEL="0"))
for name, env, se in (("frame+pipe", {}, False), ("frame nopipe", dict(CRH_PIPELINE="0"), False), ("frame pipe1", dict(CRH_FRAME_PIPE="1"), False), ("staged pipe", dict(CRH_FRAME_KERNEL="0"), False), ("frame+pipe sync every op", {}, True)):
    got = run(ctx(**env), se)
    res = []
    for a, b in zip(got, ref):
        d = bits(a) != bits(b)
        ys, xs = np.nonzero(d.any(axis=2)) if d.any() else ([], [])
        res.append(f"{int(d.any(axis=2).sum())} px" + (f" (rows {min(ys)}..{max(ys)}, cols {min(xs)}..{max(xs)})" if d.any() else ""))
    print(name, res, flush=True)
