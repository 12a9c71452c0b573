"""Hunt step 3: is a wrong ray wrong on its own?  (CRH_LIB_PATH = a failing build.)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np, torch  # noqa
from test_two_level import object_scene
from test_trace_instantiations import hunt_rays, placement
from cadrays_amd.view import View
from oracle import pyoracle
sc = object_scene(None, 128, 96)
rays, short = hunt_rays()
o = pyoracle.Oracle().load_scene(sc); o.set_transforms(placement("split")); want = o.trace_any(short)
near = o.trace_nearest(short)
v = View(0).load_scene(sc); v.set_transforms(placement("split")); v.enable_counters(True)
got = v.trace_any(short[:1024])
bad = np.nonzero(got != want[:1024])[0]
print("first 1024 rays: bad", len(bad), "of occluded", int((want[:1024] == 0).sum()), "; bad per group of 64:", np.bincount(bad // 64, minlength=16).tolist())
print("bad lanes (index mod 64) histogram:", np.bincount(bad % 64, minlength=64).tolist())
b = int(bad[0]); g = b // 64
alone = v.trace_any(short[b:b + 1])
copies = v.trace_any(np.repeat(short[b:b + 1], 64, 0))
group = v.trace_any(short[64 * g:64 * g + 64])
print(f"ray {b}: oracle {want[b]}  alone {alone.tolist()}  64 copies -> {np.unique(copies).tolist()}  in its own group of 64 -> {int(group[b - 64 * g])}; the group's bad lanes {np.nonzero(group != want[64 * g:64 * g + 64])[0].tolist()}")
# the group with the OTHER lanes replaced by rays that miss everything (pointing away, tmax tiny)
solo = short[64 * g:64 * g + 64].copy(); keep = b - 64 * g
for i in range(64):
    if i != keep: solo[i, 3] = 1e-6
r = v.trace_any(solo); print("same lane, the other 63 rays cut to tmax 1e-6 ->", int(r[keep]))
# which lanes go wrong when all 64 lanes carry THE SAME occluded ray but finish at different times?  (they all finish together: expect right)
# occluder statistics
occ = near[bad, 3].view(np.int32); print("occluders of the bad rays by object:", np.bincount(sc.tri_object[occ[occ >= 0]], minlength=7).tolist(), " all occluded rays:", np.bincount(sc.tri_object[near[:1024][want[:1024] == 0][:, 3].view(np.int32)], minlength=7).tolist())
# prefix experiment: ray b with k rays in front of it in the same wavefront
for k in (1, 2, 4, 8, 16, 32, 63):
    lo = max(64 * g, b - k)
    sub = short[lo:b + 1]
    r = v.trace_any(sub); print(f"  rays [{lo}, {b}] ({len(sub)} lanes): ray {b} -> {int(r[-1])}; wrong in this run: {np.nonzero(r != want[lo:b + 1])[0].tolist()}")
