import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
from cadrays_amd import scenes, abi
from cadrays_amd.view import View
from oracle.pyoracle import Oracle
def bits(a): return np.ascontiguousarray(a, np.float32).view(np.uint32)
from test_two_level import moved_xforms, object_scene
cases = {"cornell": (lambda: scenes.cornell_box(True, 128, 128), 6), "c3": (lambda: scenes.baseline_config("C3", 256, 144, n_tris=20000), 4),
         "c2": (lambda: scenes.baseline_config("C2", 256, 144, n_tris=20000), 4), "materials": (lambda: scenes.materials_scene(160, 120, 24, 12), 4),
         "two_level_moved": (lambda: object_scene(moved_xforms(8), 96, 96), 6), "two_level_identity": (lambda: object_scene(None, 96, 96), 6)}
for name, (mk, spp) in cases.items():
    sc = mk()
    if name == "c3": sc.env = scenes.procedural_sky(256, 128, 1)
    o = Oracle().load_scene(sc); o.render(spp); ref = o.read_hdr(); ost = o.stats(); o.close()
    v = View(0).load_scene(sc)
    t = time.time(); v.render(spp); g = v.read_hdr(); st = v.stats()
    ok = np.array_equal(bits(g), bits(ref))
    v.set_schedule(abi.SCHEDULE_STAGED); v.reset(); v.render(spp); g2 = v.read_hdr()
    print(name, "frame==oracle", ok, "staged==oracle", np.array_equal(bits(g2), bits(ref)), {k: (st[k], ost[k]) for k in ("rays_nearest", "rays_any", "shaded_hits", "samples")}, "%.2fs" % (time.time() - t), flush=True)
    v.close()
