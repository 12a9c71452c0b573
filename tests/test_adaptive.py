"""Adaptive screen sampling (reference controls AdaptiveScreenSampling and
NbRayTracingTiles at src/Launcher/SettingsWidget.cxx:427-477): each iteration renders +1 sample on tiles drawn
with probability proportional to their estimated error."""
import numpy as np
import pytest

from cadrays_amd import scenes


def adaptive_run(backend, sc, iters, tiles_per_iter):
    b = backend.load_scene(sc)
    b.set_adaptive(True, tiles_per_iter)
    b.render(iters)
    err, cnt = b.tile_stats()
    return b, err, cnt


def test_oracle_adaptive_concentrates_samples(oracle_lib):
    sc = scenes.cornell_box(True, 128, 128)                   # 16 tiles of 32x32
    o, err, cnt = adaptive_run(oracle_lib.Oracle(), sc, 40, 4)
    assert cnt.min() >= 2                                      # unsampled tiles carry error 1e3 -> everything gets seeded first
    assert cnt.sum() <= 40 * 4 and cnt.sum() >= 40 * 4 * 0.5   # duplicate picks inside an iteration collapse
    assert cnt.max() >= 2 * cnt.min()                          # and the noisy tiles (light, caustics) get visibly more
    img = o.read_hdr()
    assert np.isfinite(img).all() and img.mean() > 0.05
    # the estimate is a standard error: it shrinks as samples accumulate
    o.render(200)
    err2, cnt2 = o.tile_stats()
    assert err2.mean() < err.mean()
    # same image statistics as uniform sampling (unbiasedness of per-tile running means)
    u = oracle_lib.Oracle().load_scene(sc); u.render(24)
    assert abs(img.mean() - u.read_hdr().mean()) / u.read_hdr().mean() < 0.15


def test_oracle_adaptive_is_deterministic_and_off_restores_uniform(oracle_lib):
    sc = scenes.cornell_box(False, 64, 64)
    a, _, ca = adaptive_run(oracle_lib.Oracle(), sc, 10, 2)
    b, _, cb = adaptive_run(oracle_lib.Oracle(), sc, 10, 2)
    assert np.array_equal(a.read_hdr(), b.read_hdr()) and np.array_equal(ca, cb)
    a.set_adaptive(False, 1); a.render(3)
    u = oracle_lib.Oracle().load_scene(sc); u.render(3)
    assert np.array_equal(a.read_hdr(), u.read_hdr())


@pytest.mark.gpu
@pytest.mark.parametrize("scene_name", ["cornell", "materials"])
def test_hip_adaptive_matches_oracle_bit_exact(hip_lib, oracle_lib, scene_name):
    """identical tile choices (the per-tile error reduction has a fixed summation order), identical images."""
    from cadrays_amd.view import View
    sc = scenes.cornell_box(True, 160, 96) if scene_name == "cornell" else scenes.materials_scene(128, 96, 16, 8)
    v, ev, cv = adaptive_run(View(0), sc, 12, 5)
    o, eo, co = adaptive_run(oracle_lib.Oracle(), sc, 12, 5)
    assert np.array_equal(cv, co)
    assert np.array_equal(ev.view(np.uint32), eo.view(np.uint32))
    assert np.array_equal(v.read_hdr().view(np.uint32), o.read_hdr().view(np.uint32))


def _outlined(ldr, ts):
    """tiles whose whole 1-pixel border is pure red"""
    h, w, _ = ldr.shape
    red = (ldr[..., 0] == 255) & (ldr[..., 1] == 0) & (ldr[..., 2] == 0)
    out = []
    for ty in range((h + ts - 1) // ts):
        for tx in range((w + ts - 1) // ts):
            t = red[ty * ts:(ty + 1) * ts, tx * ts:(tx + 1) * ts]
            out.append(bool(t[0].all() and t[:, 0].all() and (t.shape[0] < ts or t[-1].all()) and (t.shape[1] < ts or t[:, -1].all())))
    return np.array(out)


def test_oracle_show_sampling_tiles_outlines_last_iteration_only(oracle_lib):
    """ShowSamplingTiles (SettingsWidget.cxx:443-449): a display overlay -- HDR read-out and accumulation are untouched."""
    sc = scenes.cornell_box(True, 128, 96)                    # 4 x 3 tiles of 32 x 32
    o, _, cnt0 = adaptive_run(oracle_lib.Oracle(), sc, 6, 3)
    plain, hdr = o.read_ldr(), o.read_hdr()
    o.set_show_tiles(True)
    assert np.array_equal(o.read_hdr(), hdr)
    o.render(1)
    _, cnt1 = o.tile_stats()
    marked = o.read_ldr()
    picked = cnt1 > cnt0
    assert 1 <= picked.sum() <= 3
    assert np.array_equal(_outlined(marked, 32), picked)
    inner = np.ones(marked.shape[:2], bool)
    for t in np.flatnonzero(picked):
        ty, tx = divmod(int(t), 4)
        inner[ty * 32:(ty + 1) * 32, tx * 32:(tx + 1) * 32] = False
        inner[ty * 32 + 1:(ty + 1) * 32 - 1, tx * 32 + 1:(tx + 1) * 32 - 1] = True
    o.set_show_tiles(False)
    assert np.array_equal(o.read_ldr()[inner], marked[inner])  # nothing but the outlines differs
    # outside adaptive mode there is nothing to show
    u = oracle_lib.Oracle().load_scene(sc); u.set_show_tiles(True); u.render(2)
    assert not _outlined(u.read_ldr(), 32).any()
    del plain


@pytest.mark.gpu
def test_hip_show_sampling_tiles_matches_oracle(hip_lib, oracle_lib):
    from cadrays_amd.view import View
    sc = scenes.cornell_box(True, 160, 96)                    # ragged last tile column
    v, _, _ = adaptive_run(View(0), sc, 9, 4)
    o, _, _ = adaptive_run(oracle_lib.Oracle(), sc, 9, 4)
    v.set_show_tiles(True); o.set_show_tiles(True)
    a, b = v.read_ldr(), o.read_ldr()
    assert np.array_equal(a, b)
    assert _outlined(a, 32).any()
    assert np.array_equal(v.read_hdr().view(np.uint32), o.read_hdr().view(np.uint32))
