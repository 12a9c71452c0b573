"""The analytic known-answer tests of tests/test_oracle_kat.py on the HIP PATH itself (formula checks must not
rest on the CPU oracle alone, which is a twin of the kernels).  Scene-level answers go through crh_render; BSDF eval / pdf /
sample / Fresnel go through the crh_debug_bsdf hook, which runs the same device functions k_shade calls."""
import dataclasses

import numpy as np
import pytest

from cadrays_amd import scenes
from cadrays_amd.materials import BSDF, Fresnel
from cadrays_amd.scenes import Camera, Light, Params, Scene

import test_oracle_kat as K

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def view(hip_lib):
    from cadrays_amd.view import View
    return View


# ---------------------------------------------------------------------------------------------- geometry (a7)
def test_single_triangle_hit_vectors_gpu(view):
    pos = np.array([[0, 0, 0], [2, 0, 0], [0, 2, 0]], np.float32)
    nrm = np.tile(np.array([[0, 0, 1]], np.float32), (3, 1))
    tri = np.array([[0, 1, 2, 0]], np.int32)
    v = view(0).load_scene(Scene(pos, nrm, tri, [BSDF.CreateDiffuse(0.5)]))
    rays = np.array([[0.5, 0.5, 3.0, 1e15, 0, 0, -1, 0], [1.0, 0.25, -2.0, 1e15, 0, 0, 1, 0], [1.5, 1.5, 1.0, 1e15, 0, 0, -1, 0],
                     [0.5, 0.5, 3.0, 2.5, 0, 0, -1, 0], [0.5, 0.5, 3.0, 1e15, 1, 0, 0, 0], [0.5, 0.5, 3.0, 1e15, 0, 0, 1, 0]], np.float32)
    h = v.trace_nearest(rays)
    assert list(h[:, 3].view(np.int32)) == [0, 0, -1, -1, -1, -1]
    np.testing.assert_allclose(h[0, :3], [3.0, 0.25, 0.25], rtol=0, atol=1e-7)
    np.testing.assert_allclose(h[1, :3], [2.0, 0.5, 0.125], rtol=0, atol=1e-7)
    assert list(v.trace_any(rays)) == [0, 0, 1, 1, 1, 1]


# ---------------------------------------------------------------------------------------------- Fresnel (a9)
def test_fresnel_limits_gpu(view):
    v = view(0)

    def fr(cos_i, f):
        b = BSDF.CreateDiffuse(0.5); b.FresnelCoat = f
        return v.debug_bsdf(3, b, np.array([[cos_i, 0, 0]], np.float32))[0]

    for n in (1.0, 1.33, 1.5, 1.62, 2.4):
        assert abs(fr(1.0, Fresnel.CreateDielectric(n))[0] - ((n - 1) / (n + 1)) ** 2) < 2e-7
        assert fr(1e-4, Fresnel.CreateDielectric(n))[0] > 0.99 or n == 1.0
    assert fr(-0.3, Fresnel.CreateDielectric(1.5))[0] == 1.0                  # total internal reflection
    f0 = (0.58, 0.42, 0.2)
    np.testing.assert_allclose(fr(1.0, Fresnel.CreateSchlick(f0)), f0, atol=1e-7)
    np.testing.assert_allclose(fr(0.0, Fresnel.CreateSchlick(f0)), 1.0, atol=1e-7)
    assert np.allclose(fr(0.3, Fresnel.CreateConstant(0.37)), 0.37)
    n, k = 0.8, 5.8
    assert abs(fr(1.0, Fresnel.CreateConductor(n, k))[0] - ((n - 1) ** 2 + k * k) / ((n + 1) ** 2 + k * k)) < 1e-5


# ---------------------------------------------------------------------------------------------- closed-form radiances (a10-a13)
def test_lambert_plane_under_constant_sky_is_rho_L_gpu(view):
    p, n, t = K.plane()
    cam, par = K.looking_down(max_depth=2, background=(0.7, 0.5, 0.25))
    rho = np.array([0.8, 0.6, 0.3], np.float32)
    v = view(0).load_scene(Scene(p, n, t, [BSDF.CreateDiffuse(rho)], camera=cam, params=par))
    v.render(3)
    img = v.read_hdr()
    np.testing.assert_allclose(img, np.broadcast_to(rho * np.array(par.background, np.float32), img.shape), rtol=3e-6)


def test_furnace_geometric_series_gpu(view):
    m = scenes._Mesh(); m.box((2, 2, 2), 0, (-1, -1, -1))
    pos, nrm, tri = m.arrays()
    rho, E, depth = 0.5, 0.75, 6
    b = BSDF.CreateDiffuse(rho); b.Le = np.array([E, E, E], np.float32)
    cam = Camera(eye=(0.1, -0.2, 0.05), dir=(0.3, 1, 0.2), up=(0, 0, 1), fovy_deg=60)
    par = Params(width=16, height=16, tile_size=8, max_depth=depth, russian_roulette=False)
    v = view(0).load_scene(Scene(pos, -nrm, tri, [b], camera=cam, params=par))
    v.render(2)
    np.testing.assert_allclose(v.read_hdr(), E * sum(rho ** k for k in range(depth)), rtol=2e-6)


def test_beer_lambert_slab_gpu(view):
    d, c, a = 0.5, 3.0, np.array([0.8, 0.5, 1.0])
    p0, n0, t0 = K.plane(z=0.0, half=1.0)
    p1, n1, t1 = K.plane(z=-d, half=1.0, up=False)
    pos = np.concatenate([p0, p1]); nrm = np.concatenate([n0, n1]); t1 = t1.copy(); t1[:, :3] += 4
    g = BSDF.CreateGlass(1.0, a, c, 1.0)
    cam = Camera(eye=(0.0, 0.0, 5.0), dir=(0, 0, -1), up=(0, 1, 0), fovy_deg=1.0)
    par = Params(width=8, height=8, tile_size=8, max_depth=4, background=(1.0, 1.0, 1.0))
    v = view(0).load_scene(Scene(pos, nrm, np.concatenate([t0, t1]), [g], camera=cam, params=par))
    v.render(1)
    np.testing.assert_allclose(v.read_hdr().reshape(-1, 3).mean(0), np.exp(-d * c * (1 - a)), rtol=3e-4)


def test_light_irradiance_gpu(view):
    p, n, t = K.plane()
    alpha, Le, rho = 0.3, 10.0, 0.6
    cam, par = K.looking_down(32, 32, max_depth=2)
    sc = Scene(p, n, t, [BSDF.CreateDiffuse(rho)], lights=[Light.directional((0, 0, -1), smoothness=alpha, intensity=Le)], camera=cam, params=par)
    v = view(0).load_scene(sc); v.render(64)
    want = rho * Le * np.sin(alpha) ** 2                                   # cone light: NEE + MIS with implicit hits
    assert abs(v.read_hdr().mean() - want) / want < 0.01
    v2 = view(0).load_scene(dataclasses.replace(sc, lights=[Light.directional((0, -0.6, -0.8), smoothness=0.0, intensity=Le)])); v2.render(1)
    np.testing.assert_allclose(v2.read_hdr(), rho / np.pi * Le * 0.8, rtol=1e-5)          # delta light: deterministic
    r, h, Le, rho = 0.2, 2.0, 30.0, 0.5                                      # sphere light seen from straight below
    cam = Camera(eye=(0.0, 0.0, 1.0), dir=(0, 0, -1), up=(0, 1, 0), fovy_deg=0.5)
    sc3 = Scene(p, n, t, [BSDF.CreateDiffuse(rho)], lights=[Light.positional((0, 0, h), smoothness=r, intensity=Le)], camera=cam,
                params=Params(width=8, height=8, tile_size=8, max_depth=2))
    v3 = view(0).load_scene(sc3); v3.render(256)
    want = rho / np.pi * Le * np.pi * (r / h) ** 2
    assert abs(v3.read_hdr().mean() - want) / want < 0.02


def test_running_mean_and_clamp_gpu(view):
    p, n, t = K.plane()
    cam, par = K.looking_down(max_depth=2, background=(4.0, 1.0, 0.5), radiance_clamp=2.0)
    v = view(0).load_scene(Scene(p, n, t, [BSDF.CreateDiffuse(1.0)], camera=cam, params=par))
    v.render(5)
    a, _ = v.save_accum()
    assert (a[..., 3] == 5).all()
    np.testing.assert_allclose(a[..., :3], np.broadcast_to(np.array([2.0, 1.0, 0.5], np.float32), a[..., :3].shape), rtol=3e-6)


# ---------------------------------------------------------------------------------------------- BSDF sampling (a10)
@pytest.mark.parametrize("name", sorted(K.PRESETS))
def test_pdf_integrates_to_one_and_energy_bounded_gpu(view, name):
    b = K.PRESETS[name]
    wo = np.array([0.4, 0.1, np.sqrt(1 - 0.17)], np.float32)
    w, dw = K.hemisphere_grid(160)
    v = view(0)
    wo_n = np.broadcast_to(wo, w.shape)
    total = v.debug_bsdf(1, b, wo_n, w)[:, 0].astype(np.float64).sum() * dw
    assert 0.93 < total < 1.03, total
    albedo = v.debug_bsdf(0, b, wo_n, w).astype(np.float64).sum(0) * dw
    assert (albedo <= 1.02).all() and (albedo > 0.02).any(), albedo


@pytest.mark.parametrize("name", sorted(K.PRESETS))
def test_sample_weight_matches_eval_over_pdf_gpu(view, name):
    b = K.PRESETS[name]
    wo = np.array([0.4, 0.1, np.sqrt(1 - 0.17)], np.float32)
    w, dw = K.hemisphere_grid(160)
    v = view(0)
    quad = v.debug_bsdf(0, b, np.broadcast_to(wo, w.shape), w).astype(np.float64).sum(0) * dw
    n = 200000
    seeds = np.zeros((n, 3), np.float32)
    seeds[:, 0] = (np.arange(n, dtype=np.uint32) * np.uint32(2654435761) + np.uint32(12345)).view(np.float32)     # any non-zero xorshift states
    out = v.debug_bsdf(2, b, np.broadcast_to(wo, (n, 3)), seeds)
    alive = (out[:, 6].astype(np.int32) & 1) != 0
    mc = (out[:, 3:6].astype(np.float64) * alive[:, None]).sum(0) / n
    np.testing.assert_allclose(mc, quad, rtol=0.03, atol=0.003)
    # and the sampled direction's density is the one pdf() reports: E[1 / pdf] over non-delta samples = measure of the support
    nd = alive & ((out[:, 6].astype(np.int32) & 2) == 0)
    pdf = v.debug_bsdf(1, b, np.broadcast_to(wo, (n, 3)), out[:, 0:3])[:, 0].astype(np.float64)
    assert (pdf[nd] > 0).all()


def test_glass_sampling_reflect_transmit_split_gpu(view):
    b = BSDF.Glass(ior=1.5)
    wo = np.array([0.0, 0.6, 0.8], np.float32)
    v = view(0)
    fb = BSDF.CreateDiffuse(0.5); fb.FresnelCoat = Fresnel.CreateDielectric(1.5)
    F = v.debug_bsdf(3, fb, np.array([[0.8, 0, 0]], np.float32))[0, 0]
    n = 100000
    seeds = np.zeros((n, 3), np.float32)
    seeds[:, 0] = (np.arange(n, dtype=np.uint32) * np.uint32(40503) + np.uint32(99)).view(np.float32)
    out = v.debug_bsdf(2, b, np.broadcast_to(wo, (n, 3)), seeds)
    fl = out[:, 6].astype(np.int32)
    assert ((fl & 1) != 0).all() and ((fl & 2) != 0).all()                       # alive, delta
    np.testing.assert_allclose(out[:, 3:6], 1.0, rtol=1e-5)                       # weight = K * F / P(lobe) = 1 for both lobes
    inside = (fl & 4) != 0
    wi = out[:, 0:3]
    assert np.allclose(np.hypot(wi[inside, 0], wi[inside, 1]), 0.6 / 1.5, atol=1e-5) and (wi[inside, 2] < 0).all()     # Snell
    np.testing.assert_allclose(wi[~inside], np.broadcast_to([0, -0.6, 0.8], wi[~inside].shape), atol=1e-6)
    assert abs((~inside).mean() - F) < 0.005


def test_debug_bsdf_equals_oracle_bitwise(view, oracle_lib):
    """the hook evaluates the same arithmetic as the oracle's unit entry points (ties the two KAT suites together)"""
    v = view(0)
    r = np.random.default_rng(1)
    for name, b in sorted(K.PRESETS.items()):
        wo = r.normal(size=(64, 3)); wo[:, 2] = np.abs(wo[:, 2]) + 0.05; wo = (wo / np.linalg.norm(wo, axis=1, keepdims=True)).astype(np.float32)
        wi = r.normal(size=(64, 3)); wi = (wi / np.linalg.norm(wi, axis=1, keepdims=True)).astype(np.float32)
        ev, pdf = v.debug_bsdf(0, b, wo, wi), v.debug_bsdf(1, b, wo, wi)[:, 0]
        for i in range(64):
            assert np.array_equal(ev[i].view(np.uint32), oracle_lib.bsdf_eval(b, wo[i], wi[i]).view(np.uint32)), (name, i)
            assert np.float32(oracle_lib.bsdf_pdf(b, wo[i], wi[i])).view(np.uint32) == pdf[i].view(np.uint32), (name, i)
