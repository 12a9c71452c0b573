"""Hunt step 4: single wrong rays -- what do the visit counters of the failing instantiation say?"""
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
o = pyoracle.Oracle().load_scene(sc); o.set_transforms(placement("split")); want = o.trace_any(short[:1024])
v = View(0).load_scene(sc); v.set_transforms(placement("split")); v.enable_counters(True)
got = v.trace_any(short[:1024])
bad = np.nonzero(got != want)[0]
good = np.nonzero((got == want) & (want == 0))[0]
def one(b, name):
    r = short[b:b + 1]
    v.reset(); ga = int(v.trace_any(r)[0]); s = v.stats(); gn, gt = s["nodes_any"], s["tris_any"]
    v.reset(); gh = v.trace_nearest(r)[0]; s = v.stats(); hn, ht = s["nodes_nearest"], s["tris_nearest"]
    o.reset(); oa = int(o.trace_any(r)[0]); s = o.stats(); on, ot = s["nodes_any"], s["tris_any"]
    o.reset(); oh = o.trace_nearest(r)[0]; s = o.stats(); ohn, oht = s["nodes_nearest"], s["tris_nearest"]
    tri = int(oh[3:4].view(np.int32)[0])
    print(f"{name} ray {b}: any gpu {ga} (nodes {gn} tris {gt}) oracle {oa} (nodes {on} tris {ot}) | nearest gpu t {gh[0]:.5f} (nodes {hn} tris {ht}) oracle t {oh[0]:.5f} tri {tri} obj {sc.tri_object[tri] if tri >= 0 else -1} (nodes {ohn} tris {oht}) | o {r[0, :3].round(3).tolist()} d {r[0, 4:7].round(3).tolist()}", flush=True)
for b in bad[:12]: one(int(b), "BAD ")
for b in good[:6]: one(int(b), "good")
