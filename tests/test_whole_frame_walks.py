"""Whole-frame agreement of the two traversal instantiations on the GPU -- hundreds of millions of rays, far more than the oracle can re-render.

The counting kernels (crh_enable_counters(1)) are the ones whose visit counters equal the oracle's in the parity tests; the timed kernels (counters
off) are the ones bench.py measures.  Here both render the SAME frames in the wide-batch schedule, batch after batch, and every word of the HDR image
must agree; sampled tiles of the same frames are checked against the oracle elsewhere (tests/test_timed_path_parity.py, bench.py's parity gate).
Round 4 used this file to vet two visit-count optimisations of the timed kernels (deferred leaf tests, camera rays seeded from the previous batch:
profiles/r4/ab_deferred_leaf_and_seed.txt) -- bit-identical here, not faster on the GPU, dropped.
Reference gate: testing/CADRays_Testing.py:226-230 compares whole images pixel by pixel."""
import numpy as np
import pytest

from cadrays_amd import abi, scenes

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _scene(kind, w, h, n):
    sc = scenes.baseline_config(kind, w, h, n_tris=n)
    if kind == "C3":
        sc.env = scenes.procedural_sky(512, 256, 1)
    return sc


@pytest.mark.parametrize("kind,n,spp", [("C3", 200_000, 64), ("C2", 100_000, 64), ("C3", 1_000_000, 32)])
def test_timed_kernels_equal_counting_kernels_on_whole_frames(hip_lib, kind, n, spp):
    from cadrays_amd.view import View
    sc = _scene(kind, 960, 540, n)
    v = View(0).load_scene(sc)
    tiles = np.arange(v.n_tiles(), dtype=np.uint32)
    v.set_schedule(abi.SCHEDULE_WIDE)
    v.enable_counters(True); v.reset()
    for b in range(3):
        v.render_tiles(tiles, b * spp, spp)
    ref = v.read_hdr(); st1 = v.stats()
    v.enable_counters(False); v.reset()
    for b in range(3):
        v.render_tiles(tiles, b * spp, spp)
    got = v.read_hdr(); st0 = v.stats()
    assert np.array_equal(bits(got), bits(ref)), f"{(bits(got) != bits(ref)).sum()} words differ"
    for k in ("rays_nearest", "rays_any", "shaded_hits", "samples"):
        assert st0[k] == st1[k]
    assert st1["nodes_nearest"] > 0 and st0["nodes_nearest"] == 0
    v.close()


def test_api_tracer_matches_oracle_on_incoherent_rays(hip_lib, oracle_lib):
    """crh_trace_nearest / crh_trace_any (k_trace_rays, counters off) on 400 k random rays with three kinds of tmax against the oracle's tracer."""
    from cadrays_amd.view import View
    sc = _scene("C2", 64, 64, 150_000)
    v = View(0).load_scene(sc); o = oracle_lib.Oracle().load_scene(sc)
    rng = np.random.default_rng(11)
    n = 400_000
    rays = np.zeros((n, 8), np.float32)
    rays[:, :3] = rng.uniform(-1.2, 1.2, (n, 3)); rays[:, 3] = rng.choice(np.array([3.0e38, 0.7, 0.2], np.float32), n)
    d = rng.normal(size=(n, 3)); rays[:, 4:7] = d / np.linalg.norm(d, axis=1, keepdims=True)
    hg, ho = v.trace_nearest(rays), o.trace_nearest(rays)
    assert np.array_equal(bits(hg), bits(ho))
    assert np.array_equal(v.trace_any(rays), o.trace_any(rays))
    v.close(); o.close()
