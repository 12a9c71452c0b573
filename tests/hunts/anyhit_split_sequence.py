"""Hunt (round 6): the wrong any-hit answers of round 4 are back, with the artefacts kept this time (profiles/failed_builds/<hash>_pl1/pl1.so): the COUNTING
any-hit tracer of a SPLIT two-level scene calls occluded rays visible when it runs after the plain instantiations (tests/test_trace_instantiations.py,
case [False-split]).  The ISA of k_trace_rays<true, true, true> in that library and in the default build is IDENTICAL (tools/disasm_build.sh) -- so the
fault is in state or timing, not in the instruction stream.  This script asks what it depends on:
   [CRH_LIB_PATH=profiles/failed_builds/..._pl1/pl1.so] python tests/hunts/anyhit_split_sequence.py"""
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
o = pyoracle.Oracle().load_scene(sc); o.set_transforms(placement("split"))
want_s, want_r = o.trace_any(short), o.trace_any(rays)
near = o.trace_nearest(short)                       # who occludes (triangle id of the nearest hit within 0.4)
tri_obj = sc.tri_object


def report(tag, got, want):
    bad = np.nonzero(got != want)[0]
    msg = f"{tag}: {len(bad)} of {len(want)} differ"
    if len(bad):
        occ = near[bad, 3].view(np.int32)
        objs = np.bincount(tri_obj[occ[occ >= 0]], minlength=7)
        msg += f"; got values {np.unique(got[bad]).tolist()}; occluder objects (count by object id) {objs.tolist()}; first {bad[:6].tolist()}; t of the occluder {near[bad[:6], 0].round(4).tolist()}"
    print(msg, flush=True)
    return len(bad)


for trial in range(int(os.environ.get("TRIALS", "4"))):
    for seq in ("plain,plain_rays,nearest,count", "count", "plain,count", "nearest,count", "plain_rays,count", "count,count,count", "plain,plain,plain,count"):
        v = View(0).load_scene(sc); v.set_transforms(placement("split"))
        n_bad = []
        for step in seq.split(","):
            if step == "plain": v.enable_counters(False); n_bad.append(report(f"  t{trial} [{seq}] plain any(short)", v.trace_any(short), want_s))
            elif step == "plain_rays": v.enable_counters(False); n_bad.append(report(f"  t{trial} [{seq}] plain any(rays)", v.trace_any(rays), want_r))
            elif step == "nearest": v.enable_counters(False); v.trace_nearest(rays); n_bad.append(0)
            elif step == "count": v.enable_counters(True); n_bad.append(report(f"  t{trial} [{seq}] COUNT any(short)", v.trace_any(short), want_s))
        v.close()
