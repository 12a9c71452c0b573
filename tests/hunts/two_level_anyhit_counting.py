"""Hunt (round 4): the API any-hit tracer of a scene whose objects have ALL been moved, counting kernels against the plain ones and the oracle.  Two
intermediate builds of round 4 (an A/B scaffold around the triangle fetch, switched OFF) returned "visible" for 7847 of 100 000 occluded rays in the
COUNTING instantiation only -- the same sources without the scaffold are right, as are the four committed states before it (profiles/r4/README.md).
Kept as the reproducer: python tests/hunts/two_level_anyhit_counting.py  [CRH_LIB_PATH=another build]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np, torch
from test_two_level import object_scene, moved_xforms, rigid
import cadrays_amd
from cadrays_amd._lib import load_library
if not hasattr(load_library(), 'crh_query_pipeline_capacity'): cadrays_amd.pipeline_capacity = lambda: (3, 4)
from cadrays_amd.view import View
from oracle import pyoracle
sc = object_scene(None, 128, 96)
for trial in range(2):
    v = View(0).load_scene(sc); v.enable_counters(bool(trial % 2)); v.reset()
    o = pyoracle.Oracle().load_scene(sc)
    allm = moved_xforms(7)
    for k in (0, 1, 2, 4): allm[k] = rigid(3.0 * (k + 1), (0, 1, 0), (0.002 * k, 0.001, -0.003 * k))
    v.set_transforms(allm); o.set_transforms(allm)
    r = np.random.default_rng(9)
    n = 100000
    org = (r.random((n, 3)) * 1.4 - 0.2).astype(np.float32)
    d = r.normal(size=(n, 3)); d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    rays = np.zeros((n, 8), np.float32); rays[:, :3] = org; rays[:, 3] = 1e15; rays[:, 4:7] = d
    short = rays.copy(); short[:, 3] = 0.4
    for rep in range(1):
        a, b = v.trace_any(short), o.trace_any(short)
        bad = np.nonzero(a != b)[0]
        hn = o.trace_nearest(rays)
        print(f"trial {trial} counters {trial % 2} rep {rep}: {len(bad)} differ", bad[:8].tolist(), "gpu", a[bad[:8]].tolist(), "oracle", b[bad[:8]].tolist(), "nearest t", hn[bad[:8], 0].tolist())
    v.close(); o.close()
