"""Diffuse texture maps + UV scale (reference: the aspect's Kd map set at
src/ImportExport/AisMesh.cxx:340-345 and by `rttexture <node> <file> -scale S T`, ImportExportPlugin.cxx:608-752)."""
import dataclasses

import numpy as np
import pytest

from cadrays_amd import scenes
from cadrays_amd.materials import BSDF


def textured_quad(tex, scale=(1.0, 1.0), kd=0.9, res=16):
    m = scenes._Mesh(); m.quad((-1, -1, 0), (1, -1, 0), (1, 1, 0), (-1, 1, 0), (0, 0, 1), 0)
    pos, nrm, tri = m.arrays()
    uv = np.array([[0, 0], [1, 0], [1, 1], [0, 1]], np.float32)
    b = BSDF.CreateDiffuse(kd); b.texture = 0; b.texture_scale = scale
    cam = scenes.Camera(eye=(0, 0, 3), dir=(0, 0, -1), up=(0, 1, 0), is_ortho=True, ortho_scale=1.0)
    par = scenes.Params(width=res, height=res, tile_size=8, max_depth=2, background=(1.0, 1.0, 1.0))
    return scenes.Scene(pos, nrm, tri, [b], uv=uv, textures=[np.asarray(tex, np.float32)], camera=cam, params=par)


def quadrant_texture():
    t = np.zeros((8, 8, 3), np.float32)
    t[:4, :4] = (1.0, 0.0, 0.0)      # image top-left  = (u < .5, v > .5)
    t[:4, 4:] = (0.0, 1.0, 0.0)      # top-right
    t[4:, :4] = (0.0, 0.0, 1.0)      # bottom-left = uv origin
    t[4:, 4:] = (1.0, 1.0, 0.0)      # bottom-right
    return t


def test_texture_orientation_and_modulation(oracle_lib):
    o = oracle_lib.Oracle().load_scene(textured_quad(quadrant_texture()))
    o.render(4)
    img = o.read_hdr()                                     # Lambert plane under a unit sky: pixel = Kd * texel
    np.testing.assert_allclose(img[3, 3], [0.9, 0, 0], atol=1e-6)        # image top-left = +y, -x -> (u<.5, v>.5)
    np.testing.assert_allclose(img[3, 12], [0, 0.9, 0], atol=1e-6)
    np.testing.assert_allclose(img[12, 3], [0, 0, 0.9], atol=1e-6)
    np.testing.assert_allclose(img[12, 12], [0.9, 0.9, 0], atol=1e-6)


def test_constant_texture_equals_scaled_kd_bitwise(oracle_lib):
    c = np.float32(0.5)
    a = oracle_lib.Oracle().load_scene(textured_quad(np.full((4, 4, 3), c, np.float32), kd=0.8)); a.render(3)
    sc = textured_quad(np.ones((4, 4, 3), np.float32), kd=0.8)
    sc = dataclasses.replace(sc, materials=[BSDF.CreateDiffuse(np.float32(0.8) * c)], textures=[])
    b = oracle_lib.Oracle().load_scene(sc); b.render(3)
    assert np.array_equal(a.read_hdr(), b.read_hdr())


def test_uv_scale_repeats(oracle_lib):
    """-scale 2 2 tiles the image 2 x 2 over the quad: the pattern of each half equals the whole unscaled one."""
    one = oracle_lib.Oracle().load_scene(textured_quad(quadrant_texture(), (1.0, 1.0), res=32)); one.render(8)
    two = oracle_lib.Oracle().load_scene(textured_quad(quadrant_texture(), (2.0, 2.0), res=32)); two.render(8)
    a, b = one.read_hdr(), two.read_hdr()
    # pixel (2, 2) of the scaled render sits deep inside the red block of the first repeat, (2, 10) in the green one
    np.testing.assert_allclose(b[2, 2], [0.9, 0, 0], atol=1e-6); np.testing.assert_allclose(b[2, 10], [0, 0.9, 0], atol=1e-6)
    np.testing.assert_allclose(b[2, 18], [0.9, 0, 0], atol=1e-6); np.testing.assert_allclose(b[18, 2], [0.9, 0, 0], atol=1e-6)
    np.testing.assert_allclose(a[4, 4], [0.9, 0, 0], atol=1e-6)


def textured_room():
    """Cornell-like scene with a textured floor and a textured (scaled) back wall next to glass / mirror objects."""
    sc = scenes.cornell_box(True, 96, 96)
    r = np.random.default_rng(3)
    tex0 = (r.random((16, 16, 3)) * 0.9 + 0.05).astype(np.float32)
    tex1 = quadrant_texture() * 0.7 + 0.2
    uv = np.zeros((len(sc.pos), 2), np.float32)
    uv[:, 0] = sc.pos[:, 0] + 0.37 * sc.pos[:, 2]; uv[:, 1] = sc.pos[:, 1] - 0.21 * sc.pos[:, 2]
    mats = [dataclasses.replace(m) for m in sc.materials]
    mats[2] = dataclasses.replace(mats[2], texture=0, texture_scale=(3.0, 2.0))      # the white walls / floor / ceiling
    mats[0] = dataclasses.replace(mats[0], texture=1)                                # red wall
    return dataclasses.replace(sc, materials=mats, uv=uv, textures=[tex0, tex1])


def test_textured_room_oracle_sane(oracle_lib):
    sc = textured_room()
    o = oracle_lib.Oracle().load_scene(sc); o.render(4)
    p = oracle_lib.Oracle().load_scene(dataclasses.replace(sc, textures=[])); p.render(4)
    a, b = o.read_hdr(), p.read_hdr()
    assert np.isfinite(a).all() and not np.array_equal(a, b) and a.mean() < b.mean()     # textures only darken Kd here


@pytest.mark.gpu
def test_textured_room_gpu_bit_exact(hip_lib, oracle_lib):
    from cadrays_amd.view import View
    sc = textured_room()
    v = View(0).load_scene(sc); v.render(4)
    o = oracle_lib.Oracle().load_scene(sc); o.render(4)
    assert np.array_equal(v.read_hdr().view(np.uint32), o.read_hdr().view(np.uint32))
    # clearing a slot restarts accumulation and falls back to the plain Kd
    v.set_texture(0, None); v.render(2)
    o2 = oracle_lib.Oracle().load_scene(dataclasses.replace(sc, textures=[None, sc.textures[1]])); o2.render(2)
    assert np.array_equal(v.read_hdr().view(np.uint32), o2.read_hdr().view(np.uint32))


# ---- RGBA textures: alpha cuts the surface out (Kd *= a, Kt = (1 - a) + a * Kt)
def alpha_quad_scene(alpha_tex, res=16):
    """textured quad (white RGB) in front of a black-ish floor, lit by a unit sky from everywhere"""
    sc = textured_quad(alpha_tex, kd=0.9, res=res)
    return sc


def test_alpha_one_is_bitwise_the_rgb_texture(oracle_lib):
    rgb = quadrant_texture() * 0.8 + 0.1
    rgba = np.concatenate([rgb, np.ones((8, 8, 1), np.float32)], 2)
    a = oracle_lib.Oracle().load_scene(textured_quad(rgb)); a.render(3)
    b = oracle_lib.Oracle().load_scene(textured_quad(rgba)); b.render(3)
    assert np.array_equal(a.read_hdr(), b.read_hdr())


def test_alpha_zero_is_invisible_and_half_alpha_mixes(oracle_lib):
    """under a unit sky: an opaque Lambert plane returns Kd, a fully cut-out one the sky itself (1), alpha 0.5 the mean"""
    def run(alpha):
        t = np.ones((4, 4, 4), np.float32); t[..., 3] = alpha
        sc = textured_quad(t, kd=0.6, res=16)
        sc = dataclasses.replace(sc, params=dataclasses.replace(sc.params, max_depth=4))
        o = oracle_lib.Oracle().load_scene(sc); o.render(64)
        return o.read_hdr()[4:12, 4:12].mean()
    assert abs(run(1.0) - 0.6) < 1e-5
    assert abs(run(0.0) - 1.0) < 1e-5
    assert abs(run(0.5) - 0.8) < 0.02                      # 0.5 * Kd + 0.5 * transmitted sky, stochastic lobe choice


def alpha_room():
    sc = textured_room()
    r = np.random.default_rng(5)
    t = sc.textures[0]
    a = (r.random(t.shape[:2] + (1,)) > 0.4).astype(np.float32) * 0.75 + 0.25 * r.random(t.shape[:2] + (1,)).astype(np.float32)
    return dataclasses.replace(sc, textures=[np.concatenate([t, a.astype(np.float32)], 2), sc.textures[1]])


def test_set_texture_rejects_bad_channel_count(oracle_lib):
    o = oracle_lib.Oracle()
    with pytest.raises(Exception):
        o.set_texture(0, np.ones((4, 4, 2), np.float32))


@pytest.mark.gpu
def test_alpha_room_gpu_bit_exact(hip_lib, oracle_lib):
    from cadrays_amd.view import View
    sc = alpha_room()
    v = View(0).load_scene(sc); v.render(4)
    o = oracle_lib.Oracle().load_scene(sc); o.render(4)
    a, b = v.read_hdr(), o.read_hdr()
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    p = oracle_lib.Oracle().load_scene(textured_room()); p.render(4)
    assert not np.array_equal(b, p.read_hdr())              # the alpha channel did change the light transport
    with pytest.raises(Exception):
        v.set_texture(0, np.ones((4, 4, 2), np.float32))


# ---- texture coordinates far beyond the float -> int range (uv * scale >= 2^31 indexed texels out of bounds)
def huge_uv_scene(scale):
    sc = textured_quad(quadrant_texture(), scale=scale, res=16)
    uv = sc.uv.copy(); uv[1:3, 0] = 3.0e9; uv[2:, 1] = -7.5e12          # finite, so the boundary accepts them
    return dataclasses.replace(sc, uv=uv)


@pytest.mark.parametrize("scale", [(1.0, 1.0), (1.0e30, 3.0e38), (1.0e-3, 65536.0)])
def test_huge_texture_coordinates_stay_in_bounds_oracle(oracle_lib, scale):
    o = oracle_lib.Oracle().load_scene(huge_uv_scene(scale)); o.render(2)
    img = o.read_hdr()
    assert np.isfinite(img).all() and img.max() <= 0.9 + 1e-6              # every sample is Kd x a texel of the image


@pytest.mark.gpu
@pytest.mark.parametrize("scale", [(1.0, 1.0), (1.0e30, 3.0e38), (1.0e-3, 65536.0)])
def test_huge_texture_coordinates_gpu_bit_exact(hip_lib, oracle_lib, scale):
    from cadrays_amd.view import View
    sc = huge_uv_scene(scale)
    v = View(0).load_scene(sc); v.render(2)
    o = oracle_lib.Oracle().load_scene(sc); o.render(2)
    assert np.array_equal(v.read_hdr().view(np.uint32), o.read_hdr().view(np.uint32))
