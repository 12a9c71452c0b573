"""crh_build_prebuilt: one process per GPU on one host builds the BVH ONCE (the scene is replicated) -- rank 0's tree, exported with
crh_get_bvh, is handed to the other contexts instead of being built again (bench.py --gpus N, cadrays_amd/sharding.py load_scene_shared).  The
context that takes it must render the same bits, hold the same bytes, and refuse anything that is not a tree over its own geometry.
Reference: every OCCT view builds its own BVH behind AIS_InteractiveContext::Display (AisMesh.cxx:357-423)."""
import dataclasses

import numpy as np
import pytest

from cadrays_amd import scenes

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def test_prebuilt_context_is_the_built_one(hip_lib, oracle_lib):
    from cadrays_amd.view import View
    sc = scenes.baseline_config("C3", 320, 200, n_tris=60_000)
    sc.env = scenes.procedural_sky(256, 128, 1)
    a = View(0).load_scene(sc)
    nodes, order = a.export_tree()
    assert sorted(order.tolist()) == list(range(len(sc.tri)))
    b = View(0).load_scene(sc, prebuilt=(nodes, order))
    na, ta = a.get_bvh(); nb, tb = b.get_bvh()
    assert np.array_equal(na.view(np.uint32), nb.view(np.uint32)) and np.array_equal(ta.view(np.uint32), tb.view(np.uint32))
    a.enable_counters(True); b.enable_counters(True); a.reset(); b.reset()
    a.render(3); b.render(3)
    assert np.array_equal(bits(a.read_hdr()), bits(b.read_hdr())) and a.stats() | {"seconds": 0} == b.stats() | {"seconds": 0}
    o = oracle_lib.Oracle().load_scene(sc); o.render(3)
    assert np.array_equal(bits(b.read_hdr()), bits(o.read_hdr()))
    # the API tracers see the same tree
    rng = np.random.default_rng(3)
    rays = np.zeros((20000, 8), np.float32); rays[:, :3] = rng.uniform(-1.2, 1.2, (20000, 3)); rays[:, 3] = 3.0e38
    d = rng.normal(size=(20000, 3)); rays[:, 4:7] = d / np.linalg.norm(d, axis=1, keepdims=True)
    assert np.array_equal(bits(a.trace_nearest(rays)), bits(b.trace_nearest(rays)))
    a.close(); b.close(); o.close()


def test_prebuilt_tree_is_validated(hip_lib):
    from cadrays_amd import abi
    from cadrays_amd.binding import BackendError
    from cadrays_amd.view import View
    sc = scenes.baseline_config("C2", 64, 64, n_tris=2_000)
    a = View(0).load_scene(sc)
    nodes, order = a.export_tree()
    v = View(0)

    def attempt(nd, od, scene=sc):
        with pytest.raises(BackendError):
            v.load_scene(scene, prebuilt=(nd, od))
    attempt(nodes, order[:-1])                                        # wrong triangle count
    bad = order.copy(); bad[5] = bad[6]
    attempt(nodes, bad)                                               # not a permutation
    bad = order.copy(); bad[0] = len(order)
    attempt(nodes, bad)                                               # triangle index out of range
    nb = nodes.copy().view(np.uint32); nb[0, 10] = len(nodes) + 7
    attempt(nb.view(np.float32), order)                               # child block outside the node array
    nb = nodes.copy().view(np.uint32); nb[0, 10] = 0
    attempt(nb.view(np.float32), order)                               # child block that points back (a cycle)
    nb = nodes.copy().view(np.uint32)
    leafy = [i for i in range(len(nb)) if (nb[i, 3] >> 28) & 7 > (nb[i, 3] >> 24) & 7][0]
    nb[leafy, 11] = 0x80000000 | len(order)
    attempt(nb.view(np.float32), order)                               # leaf positions beyond the triangles
    nb = nodes.copy().view(np.uint32); nb[0, 3] |= 7 << 28
    attempt(nb.view(np.float32), order)                               # seven children
    attempt(nodes[:0], order)                                         # no nodes
    # a child box that does not hold what hangs below it (silently wrong images) ...
    nb = nodes.copy().view(np.uint32)
    wide = [i for i in range(len(nb)) if (nb[i, 3] >> 28) & 7 >= 1 and any(((nb[i, 7 + x] >> 0) & 0xff) - ((nb[i, 4 + x] >> 0) & 0xff) >= 4 for x in range(3))][0]
    ax = [x for x in range(3) if ((nb[wide, 7 + x]) & 0xff) - ((nb[wide, 4 + x]) & 0xff) >= 4][0]
    nb[wide, 7 + ax] = (nb[wide, 7 + ax] & ~np.uint32(0xff)) | (((nb[wide, 7 + ax] & 0xff) - 3) & 0xff)      # slot 0's upper plane pulled in by three grid steps
    attempt(nb.view(np.float32), order)
    # ... and a tree deeper than the traversal stack (kLdsStack + kOvfStack = 128 pending entries): a forward chain of 4-wide nodes, three leaves and one inner
    # child per level, every box spanning the scene (the planes 0 .. 255 of the real root's grid)
    def chain(levels):
        scn = scenes.baseline_config("C2", 64, 64, n_tris=3 * (levels - 1) + 4)
        b = View(0).load_scene(scn); root = b.export_tree()[0].view(np.uint32)[0].copy(); b.close()
        cn = np.zeros((levels, abi.NODE_DWORDS), np.uint32)
        for lvl in range(levels):
            last = lvl == levels - 1
            cn[lvl, :3] = root[:3]
            cn[lvl, 3] = (root[3] & 0x00ffffff) | ((0 if last else 1) << 24) | (4 << 28)
            cn[lvl, 4:7] = 0; cn[lvl, 7:10] = 0xffffffff
            cn[lvl, 10] = 0 if last else lvl + 1
            cn[lvl, 11] = 0x80000000 | ((3 * lvl) if last else (3 * lvl))      # slots: [inner child,] three (last level: four) consecutive leaves
        return scn, cn.view(np.float32), np.arange(3 * (levels - 1) + 4, dtype=np.uint32)
    scn, cn, od = chain(45)                                           # 44 levels x 3 pending siblings + 3 > 128
    with pytest.raises(BackendError, match="deeper than the traversal stack"):
        v.load_scene(scn, prebuilt=(cn, od))
    scn, cn, od = chain(40)                                           # 39 x 3 + 3 = 120: fits, and renders what its own tree renders
    w = View(0).load_scene(scn, prebuilt=(cn, od)); w.render(2)
    own = View(0).load_scene(scn); own.render(2)
    assert np.array_equal(bits(w.read_hdr()), bits(own.read_hdr()))
    w.close(); own.close()
    two = scenes.baseline_config("C2", 64, 64, n_tris=2_000)
    two = dataclasses.replace(two, tri_object=np.zeros(len(two.tri), np.int32), obj_xform=np.array([[1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0]], np.float32))
    attempt(nodes, order, two)                                        # two-level scenes build their own trees
    v.load_scene(sc, prebuilt=(nodes, order))                         # and the context is still usable afterwards
    v.render(1); a.reset(); a.render(1)
    assert np.array_equal(bits(v.read_hdr()), bits(a.read_hdr()))
    v.close(); a.close()


def test_deeper_pipeline_than_the_hardware_queues_is_refused():
    """crh_set_pipeline_depth against crh_query_pipeline_capacity (verdict r3 item 8) in a process the host left at the runtime's default four queues"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import torch, cadrays_amd\n"
            "from cadrays_amd import scenes\n"
            "from cadrays_amd.view import View\n"
            "from cadrays_amd.binding import BackendError\n"
            "assert cadrays_amd.pipeline_capacity() == (3, 4)\n"
            "v = View(0).load_scene(scenes.cornell_box(True, 64, 64))\n"
            "v.set_pipeline_depth(3); v.set_pipeline_depth(2)\n"
            "try:\n    v.set_pipeline_depth(5); raise SystemExit('accepted')\n"
            "except BackendError as e:\n    assert 'GPU_MAX_HW_QUEUES' in str(e) and 'crh_query_pipeline_capacity' in str(e), str(e)\n"
            "v.render(2); print('ok')\n")
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    p = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and p.stdout.strip().endswith("ok"), (p.stdout + p.stderr)[-2000:]
