"""k_trace_packets: the camera rays of a wide batch walk the tree as one packet per wavefront (64 samples of one pixel: one shared stack with lane masks, scalar
node / triangle fetches).  The hit of a ray must not depend on that -- except among triangles at EXACTLY the same distance, where the spec says "first in the
ray's own near-to-far walk"; a lane that meets such a tie leaves the packet's result alone and goes through the per-ray kernel afterwards (the fall-back pass).
  * packets on / off (CRH_PACKETS): identical whole frames, on scenes with shared edges (Cornell: ties happen) and on a triangle soup;
  * a scene built to tie EVERYWHERE (every triangle twice, bit for bit) -- every camera ray takes the fall-back pass -- still equals the oracle;
  * partial packets (image edges, tile subsets, 16 / 48 / 80 samples per batch) and the C1 configuration at its real size against the oracle.
Reference: the reference's gate compares whole images pixel by pixel (testing/CADRays_Testing.py:226-230)."""
import dataclasses

import numpy as np
import pytest

from cadrays_amd import abi, scenes

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def render_wide(sc, spp, batches, packets, monkeypatch, tiles=None):
    from cadrays_amd.view import View
    monkeypatch.setenv("CRH_PACKETS", "16" if packets else "0")      # 16: packets also where a wavefront holds 2 - 4 pixels (the default threshold is 64)
    v = View(0).load_scene(sc)
    v.set_schedule(abi.SCHEDULE_WIDE)
    t = np.arange(v.n_tiles(), dtype=np.uint32) if tiles is None else tiles
    for b in range(batches):
        v.render_tiles(t, b * spp, spp)
    img, st = v.read_hdr(), v.stats()
    v.close()
    return img, st


@pytest.mark.parametrize("name,spp", [("cornell", 64), ("cornell_odd_size", 48), ("materials", 32), ("soup", 16), ("soup", 80)])
def test_packets_do_not_change_a_bit(hip_lib, monkeypatch, name, spp):
    sc = {"cornell": lambda: scenes.cornell_box(True, 256, 256), "cornell_odd_size": lambda: scenes.cornell_box(True, 203, 131),
          "materials": lambda: scenes.materials_scene(320, 240, 24, 12), "soup": lambda: scenes.baseline_config("C2", 416, 300, n_tris=60_000)}[name]()
    a, sa = render_wide(sc, spp, 2, True, monkeypatch)
    b, sb = render_wide(sc, spp, 2, False, monkeypatch)
    assert np.array_equal(bits(a), bits(b)), f"{(bits(a) != bits(b)).sum()} words differ"
    for k in ("rays_nearest", "rays_any", "shaded_hits", "samples"):
        assert sa[k] == sb[k]


def test_every_camera_ray_in_the_fall_back_pass(hip_lib, oracle_lib, monkeypatch):
    """every triangle of the Cornell box TWICE (the copy appended: same vertices, same bits): each camera ray that hits anything meets two triangles at
    exactly the same distance, the walk order decides which one it reports -- packets must hand every such ray to the per-ray pass"""
    sc = scenes.cornell_box(True, 160, 128)
    tri2 = np.concatenate([sc.tri, sc.tri[::-1]])              # the copies in reverse order: another builder order than the originals
    sc2 = dataclasses.replace(sc, tri=tri2)
    o = oracle_lib.Oracle().load_scene(sc2); o.render(32)
    ref = o.read_hdr(); o.close()
    img, _ = render_wide(sc2, 32, 1, True, monkeypatch)
    assert np.array_equal(bits(img), bits(ref))
    img0, _ = render_wide(sc2, 32, 1, False, monkeypatch)
    assert np.array_equal(bits(img0), bits(ref))


def test_partial_packets_and_tile_subsets_against_the_oracle(hip_lib, oracle_lib, monkeypatch):
    sc = scenes.baseline_config("C3", 300, 170, n_tris=30_000)      # 300 x 170 in 32 x 32 tiles: partial tiles right and bottom
    sc.env = scenes.procedural_sky(256, 128, 1)
    o = oracle_lib.Oracle().load_scene(sc)
    sub = np.array([0, 3, 9, 10, 17, 29, 41, 50, 59], np.uint32)     # includes edge tiles (column 9, row 5)
    o.render_tiles(sub, 16, 48)
    ref = o.read_accum(); o.close()
    img, _ = render_wide(sc, 48, 1, True, monkeypatch, tiles=sub)
    from cadrays_amd.view import View
    monkeypatch.setenv("CRH_PACKETS", "16")
    v = View(0).load_scene(sc); v.set_schedule(abi.SCHEDULE_WIDE); v.render_tiles(sub, 16, 48)
    g = v.read_hdr(); v.close()
    mask = ref[..., 3] == 48
    assert mask.sum() > 5000 and np.array_equal(bits(g[mask]), bits(ref[..., :3][mask]))


# the packet walk is instantiated once per direction octant (the byte words of a node are picked by register choice) plus once with per-lane signs for the
# packets that straddle an axis of the view: every one of the nine must give the per-ray walk's hits
_OCTANT_VIEWS = [(sx, sy, sz, 20.0) for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)] + [(1, 0, 0, 70.0), (0, -1, 0, 70.0), (0, 0, 1, 70.0)]


@pytest.mark.parametrize("dx,dy,dz,fov", _OCTANT_VIEWS)
def test_every_direction_octant_and_the_axis_straddling_packets(hip_lib, oracle_lib, monkeypatch, dx, dy, dz, fov):
    sc = scenes.baseline_config("C2", 192, 160, n_tris=40_000)
    d = np.array([dx, dy, dz], np.float64); d /= np.linalg.norm(d)
    up = (0.0, 0.0, 1.0) if abs(d[2]) < 0.9 else (0.0, 1.0, 0.0)
    # narrow views from outside along a diagonal: every pixel's directions share the signs (dx, dy, dz); wide views along an axis: the centre packets mix them
    sc = dataclasses.replace(sc, camera=dataclasses.replace(sc.camera, eye=tuple(-3.2 * d), dir=tuple(d), up=up, fovy_deg=fov))
    a, sa = render_wide(sc, 64, 1, True, monkeypatch)
    b, sb = render_wide(sc, 64, 1, False, monkeypatch)
    assert sa["rays_nearest"] == sb["rays_nearest"] and sa["shaded_hits"] == sb["shaded_hits"] and sa["shaded_hits"] > 50_000
    assert np.array_equal(bits(a), bits(b)), f"{(bits(a) != bits(b)).sum()} words differ"
    if (dx, dy, dz) in ((1, 1, 1), (-1, 1, -1), (0, -1, 0)):
        o = oracle_lib.Oracle().load_scene(sc); o.render(64)
        ref = o.read_hdr(); o.close()
        assert np.array_equal(bits(a), bits(ref))


def test_a_scene_of_objects_walks_the_64_byte_nodes(hip_lib, oracle_lib, monkeypatch):
    """single-level scenes get a second node array for the packets (the quantised planes as floats, k_expand_packet_nodes); a scene of placed objects does not
    (its node array grows while objects are dragged) and its packets read the 64-byte nodes: the other instantiation of k_trace_packets.  Both give the oracle's frame."""
    from test_two_level import object_scene
    sc = object_scene(None, 160, 128)                            # every object at its build-time placement: rendered by the single-level kernels
    o = oracle_lib.Oracle().load_scene(sc); o.render(64)
    ref = o.read_hdr(); o.close()
    a, _ = render_wide(sc, 64, 1, True, monkeypatch)
    b, _ = render_wide(sc, 64, 1, False, monkeypatch)
    assert np.array_equal(bits(a), bits(ref)) and np.array_equal(bits(b), bits(ref))
    flat = dataclasses.replace(sc, tri_object=None, obj_xform=None)      # the same triangles without objects: the packet-node array
    c, _ = render_wide(flat, 64, 1, True, monkeypatch)
    assert np.array_equal(bits(c), bits(ref))
