"""Parity of the kernels and the schedule the HEADLINE is measured on (round-2 verdict, "What's weak" 2).

bench.py times the big-batch schedule: one stream, full persistent grids, `k_trace_nearest<COUNT=false, TWO=false, DON=false>`
(`crh_api.cpp` run_batch).  The other render-parity tests either switch the visit counters on (-> the COUNT=true instantiations) or
render batches small enough for the small-batch schedule (two tile ranges / pipelined frames, the work-DONATING instantiations).  Here
every scene goes through `crh_set_schedule(CRH_SCHEDULE_WIDE)` with the counters OFF -- exactly the timed instantiation and launch
order -- and, separately, through CRH_SCHEDULE_SMALL (round 5: the frame kernel, one launch per batch) and CRH_SCHEDULE_STAGED (the small-batch
schedule of rounds 2 - 4: the donating kernels) with the counters off, each against the oracle, not only against another GPU run, and the HDR image
must equal the CPU oracle's bit for bit.  C1 runs at its real size (BASELINE.json configs[0]:
512 x 512, 64 spp, depth 5).  The reference's own gate is pixel-exact as well (testing/CADRays_Testing.py:226-230)."""
import dataclasses

import numpy as np
import pytest

from cadrays_amd import abi, scenes

from test_two_level import moved_xforms, object_scene

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _c3_small():
    sc = scenes.baseline_config("C3", 256, 144, n_tris=20_000)
    sc.env = scenes.procedural_sky(256, 128, 1)
    return sc


CASES = {
    "cornell_full": (lambda: scenes.cornell_box(True, 128, 128), 8),
    "materials": (lambda: scenes.materials_scene(160, 120, 24, 12), 4),
    "c2_small": (lambda: scenes.baseline_config("C2", 256, 144, n_tris=20_000), 4),
    "c3_small": (_c3_small, 4),
    "two_level_moved": (lambda: object_scene(moved_xforms(8), 96, 96), 6),
    "two_level_identity": (lambda: object_scene(None, 96, 96), 6),
}

_oracle_cache = {}


def oracle_image(oracle_lib, name):
    if name not in _oracle_cache:
        mk, spp = CASES[name]
        o = oracle_lib.Oracle().load_scene(mk()); o.render(spp)
        _oracle_cache[name] = (o.read_hdr(), o.read_ldr(), o.stats())
        o.close()
    return _oracle_cache[name]


@pytest.mark.parametrize("schedule", ["wide", "small", "staged"])
@pytest.mark.parametrize("name", sorted(CASES))
def test_timed_instantiation_matches_oracle(hip_lib, oracle_lib, name, schedule):
    from cadrays_amd.view import View
    mk, spp = CASES[name]
    ref_hdr, ref_ldr, ref_st = oracle_image(oracle_lib, name)
    v = View(0).load_scene(mk())
    # small: the frame kernel (one launch per batch, k_frame.h; round 5); staged: the small-batch schedule of rounds 2 - 4 (launches per stage and bounce)
    v.set_schedule({"wide": abi.SCHEDULE_WIDE, "small": abi.SCHEDULE_SMALL, "staged": abi.SCHEDULE_STAGED}[schedule])
    v.enable_counters(False); v.reset()
    v.render(spp)
    g = v.read_hdr()
    assert np.array_equal(bits(g), bits(ref_hdr)), f"{name}/{schedule}: counters-off image differs from the oracle"
    assert np.array_equal(v.read_ldr(), ref_ldr)
    st = v.stats()
    for k in ("rays_nearest", "rays_any", "shaded_hits", "samples"):       # the ray counts are kept by every instantiation
        assert st[k] == ref_st[k], (k, st[k], ref_st[k])
    assert st["nodes_nearest"] == 0 and st["tris_nearest"] == 0            # proof that the counting kernels did NOT run
    # and frame by frame (what Redraw() does), same schedule
    v.reset()
    for _ in range(spp):
        v.Redraw()
    assert np.array_equal(bits(v.read_hdr()), bits(ref_hdr))
    v.close()


def test_schedule_switch_validates_and_restores(hip_lib):
    from cadrays_amd.binding import BackendError
    from cadrays_amd.view import View
    sc = scenes.cornell_box(True, 64, 64)
    v = View(0).load_scene(sc)
    with pytest.raises(BackendError):
        v.set_schedule(7)
    v.render(3); a = v.read_hdr()
    for mode in (abi.SCHEDULE_WIDE, abi.SCHEDULE_SMALL, abi.SCHEDULE_STAGED, abi.SCHEDULE_AUTO):
        v.set_schedule(mode); v.reset(); v.render(3)
        assert np.array_equal(bits(v.read_hdr()), bits(a))
    v.close()


def test_c1_at_its_real_size_bit_exact(hip_lib, oracle_lib):
    """BASELINE.json configs[0] as written: Cornell box (34 triangles, diffuse), 512 x 512, 64 spp, depth 5 -- whole image, counters
    included (second pass), against the oracle."""
    from cadrays_amd.view import View
    sc = scenes.baseline_config("C1")
    assert (sc.params.width, sc.params.height, sc.params.max_depth) == (512, 512, 5) and len(sc.tri) == 34
    o = oracle_lib.Oracle().load_scene(sc); o.render(64)
    ref, ost = o.read_hdr(), o.stats()
    v = View(0).load_scene(sc)
    v.render(64)                                        # 16.8 M paths: the wide schedule, plain kernels
    g = v.read_hdr()
    rel = float(np.linalg.norm(g.astype(np.float64) - ref) / np.linalg.norm(ref.astype(np.float64)))
    assert rel <= 1e-4, rel
    assert np.array_equal(bits(g), bits(ref))
    assert np.array_equal(v.read_ldr(), o.read_ldr())
    v.enable_counters(True); v.reset(); v.render(64)
    st = v.stats()
    for k in ("rays_nearest", "rays_any", "nodes_nearest", "tris_nearest", "nodes_any", "tris_any", "shaded_hits", "samples"):
        assert st[k] == ost[k], (k, st[k], ost[k])
    assert np.array_equal(bits(v.read_hdr()), bits(ref))
    v.close(); o.close()


@pytest.mark.parametrize("spp", [128, 512])
def test_wide_batch_of_the_bench_shape_matches_oracle_on_tiles(hip_lib, oracle_lib, spp):
    """The shape bench.py times -- one crh_render_tiles call over every tile of a 1080p frame, one pixel x 64 samples per wavefront: 128 spp (one batch:
    rounds 2 - 3) and 512 spp (round 4: two tile groups of 1024 tiles x 512 samples, crh_schedule.cpp) -- on a 1080p C3 with a smaller soup (the oracle
    builds its tree in a second): sampled tiles, the last of the first group and the first of the second among them, bit-exact."""
    from cadrays_amd.view import View
    sc = scenes.baseline_config("C3", n_tris=50_000)
    sc.env = scenes.procedural_sky(512, 256, 1)
    v = View(0).load_scene(sc)
    tiles = np.arange(v.n_tiles(), dtype=np.uint32)
    v.render_tiles(tiles, spp, spp)                    # samples spp .. 2 spp - 1, as bench.py's first timed step after one warm-up step
    g = v.read_hdr()
    o = oracle_lib.Oracle().load_scene(sc)
    sample = np.unique(np.concatenate([np.linspace(0, o.n_tiles() - 1, 6 if spp == 128 else 3), [1023, 1024] if spp == 512 else []]).astype(np.uint32))
    o.render_tiles(sample, spp, spp)
    ref = o.read_accum()
    mask = ref[..., 3] == spp
    assert mask.sum() >= (len(sample) - 1) * 32 * 32
    assert np.array_equal(bits(g[mask]), bits(ref[..., :3][mask]))
    v.close(); o.close()


@pytest.mark.parametrize("budget_tiles,spp", [(5, 320), (3, 200), (70, 320), (16, 700), (300, 96)])
def test_tile_groups_and_sample_batches_do_not_change_a_bit(hip_lib, oracle_lib, budget_tiles, spp):
    """a call that does not fit one batch is cut into tile groups of up to 1024 samples (multiples of 64) and the rest of the samples (crh_schedule.cpp):
    whatever the cut -- path budgets that hold 3 .. 300 tiles x 64 samples, sample counts that are not multiples of 64 -- the frame is the oracle's"""
    from cadrays_amd.view import View
    sc = scenes.baseline_config("C3", 208, 150, n_tris=20_000)          # 7 x 5 tiles, partial ones right and bottom
    sc.env = scenes.procedural_sky(256, 128, 1)
    o = oracle_lib.Oracle().load_scene(sc); o.render(spp)
    ref = o.read_hdr(); o.close()
    v = View(0).load_scene(sc)
    v.set_schedule(abi.SCHEDULE_WIDE)
    v.set_path_budget(budget_tiles * 1024 * 64)
    v.render_tiles(np.arange(v.n_tiles(), dtype=np.uint32), 0, spp)
    g = v.read_hdr(); v.close()
    assert np.array_equal(bits(g), bits(ref))
