"""Randomised scenes through both sides of the boundary: every Fresnel model, coats, absorption, emission, textures with alpha,
several lights of both kinds, pinhole / thin-lens / orthographic cameras, object transforms, odd image sizes and every
rendering switch, drawn from a seeded generator.  The HIP path must reproduce the oracle bit for bit (image, LDR read-out
and traversal counters) on each of them."""
import numpy as np
import pytest

from cadrays_amd import scenes
from cadrays_amd.materials import BSDF, Fresnel
from cadrays_amd.scenes import Camera, Light, Params, Scene

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def view_cls(hip_lib):
    import torch  # noqa: F401  (runtime ordering: torch's HIP runtime first)
    from cadrays_amd.view import View
    return View


@pytest.fixture(scope="module")
def Oracle(oracle_lib):
    return oracle_lib.Oracle


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def random_fresnel(r):
    k = r.integers(0, 4)
    if k == 0:
        return Fresnel.CreateConstant(r.random())
    if k == 1:
        return Fresnel.CreateSchlick(r.random(3))
    if k == 2:
        return Fresnel.CreateConductor(0.1 + 3 * r.random(), 0.1 + 5 * r.random())
    return Fresnel.CreateDielectric(1.0 + 1.5 * r.random())


def random_bsdf(r, n_tex):
    b = BSDF()
    on = r.random(5) < 0.6
    if on[0]:
        b.Kd = r.random(3).astype(np.float32)
    if on[1]:
        b.Ks = np.append(r.random(3), r.choice([0.0, 0.02, 0.1, 0.5, 1.0])).astype(np.float32)
    if on[2]:
        b.Kt = r.random(3).astype(np.float32)
    if on[3]:
        b.Kc = np.append(r.random(3), r.choice([0.0, 0.05, 0.3])).astype(np.float32)
    if on[4] and r.random() < 0.3:
        b.Le = (2 * r.random(3)).astype(np.float32)
    b.FresnelBase, b.FresnelCoat = random_fresnel(r), random_fresnel(r)
    if r.random() < 0.5:
        b.Absorption = np.append(r.random(3), 4 * r.random()).astype(np.float32)
    if n_tex and r.random() < 0.5:
        b.texture = int(r.integers(0, n_tex)); b.texture_scale = (float(r.choice([1.0, 2.0, 0.5])), float(r.choice([1.0, 3.0])))
    return b.Sanitize()


def random_scene(seed):
    r = np.random.default_rng(seed)
    n = int(r.choice([1, 7, 60, 400, 2500]))
    n_mat = int(r.integers(1, 7))
    pos, nrm, tri = scenes.gen_scene(n, int(r.integers(1, 1000)), n_mat)
    pos = (pos * np.float32(r.choice([1.0, 0.01, 50.0]))).astype(np.float32)        # scene scale: epsilon handling
    scale = float(np.abs(pos).max())
    if r.random() < 0.5:                                                             # smooth-ish normals instead of face normals
        nrm = (nrm + 0.3 * r.normal(size=nrm.shape)).astype(np.float32)
        nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    n_tex = int(r.integers(0, 3))
    textures = []
    for _ in range(n_tex):
        h, w, ch = int(r.integers(1, 9)), int(r.integers(1, 9)), int(r.choice([3, 4]))
        textures.append(r.random((h, w, ch)).astype(np.float32))
    uv = (3 * r.random((len(pos), 2)) - 1).astype(np.float32) if n_tex else None
    mats = [random_bsdf(r, n_tex) for _ in range(n_mat)]
    lights = []
    for _ in range(int(r.integers(0, 4))):
        if r.random() < 0.5:
            lights.append(Light.directional(r.normal(size=3), float(r.choice([0.0, 0.05, 0.4])), 1 + 5 * r.random(), r.random(3)))
        else:
            lights.append(Light.positional(scale * (2 * r.random(3) - 1), float(r.choice([0.0, 0.05, 0.3])) * scale, (1 + 10 * r.random()) * scale * scale, r.random(3)))
    env = scenes.procedural_sky(32, 16, int(seed)) if r.random() < 0.5 else None
    eye = np.array([0.0, -3.6 * scale, 0.3 * scale]) + 0.2 * scale * r.normal(size=3)
    cam = Camera(eye=tuple(eye), dir=tuple(-eye / np.linalg.norm(eye)), up=(0.0, 0.0, 1.0), fovy_deg=float(r.choice([30.0, 45.0, 90.0])))
    mode = r.integers(0, 3)
    if mode == 1:
        cam.is_ortho = True; cam.ortho_scale = 1.2 * scale
    elif mode == 2:
        cam.aperture_radius = 0.05 * scale; cam.focal_dist = 3.0 * scale
    par = Params(width=int(r.integers(9, 70)), height=int(r.integers(9, 50)), max_depth=int(r.choice([1, 2, 5, 12, 32])),
                 radiance_clamp=float(r.choice([0.0, 1.0, 30.0])), two_sided=bool(r.integers(0, 2)), coherent_rng=bool(r.integers(0, 2)),
                 seed=int(r.integers(1, 1 << 30)), tile_size=int(r.choice([8, 16, 32, 64])), tonemap_mode=int(r.integers(0, 2)),
                 exposure=float(r.choice([0.0, -1.0, 2.0])), white_point=float(r.choice([1.0, 4.0])), background=tuple(r.random(3)),
                 env_as_background=bool(r.integers(0, 2)), russian_roulette=bool(r.integers(0, 2)))
    sc = Scene(pos, nrm, tri, mats, lights, env, cam, par, uv=uv, textures=textures, name=f"fuzz{seed}")
    if r.random() < 0.4 and n >= 7:                                                  # two-level: objects with transforms
        n_obj = int(r.integers(1, 5))
        sc.tri_object = (np.arange(n) % n_obj).astype(np.int32)
        # every vertex must belong to one object: the generator emits 3 private vertices per triangle
        xf = np.zeros((n_obj, 12), np.float32)
        for o in range(n_obj):
            a = r.random() * 6.28; s = float(r.choice([1.0, 0.7, 1.5]))
            m = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]]) * s
            xf[o] = np.concatenate([m, (0.2 * scale * r.normal(size=3))[:, None]], axis=1).reshape(-1)
        sc.obj_xform = xf
    return sc


@pytest.mark.parametrize("seed", range(1, 97))
def test_random_scene_bit_exact(view_cls, Oracle, seed):
    sc = random_scene(seed)
    spp = 1 + seed % 3
    v = view_cls(0).load_scene(sc); v.enable_counters(True); v.reset()
    o = Oracle().load_scene(sc)
    v.render(spp); o.render(spp)
    g, c = v.read_hdr(), o.read_hdr()
    assert np.array_equal(bits(g), bits(c)), (seed, float(np.abs(g - c).max()))
    assert np.array_equal(v.read_ldr(), o.read_ldr())
    gs, cs = v.stats(), o.stats()
    for k in ("rays_nearest", "rays_any", "nodes_nearest", "tris_nearest", "nodes_any", "tris_any", "shaded_hits", "samples"):
        assert gs[k] == cs[k], (seed, k, gs[k], cs[k])
    if sc.tri_object is not None:
        # objects placed at build time are baked into one tree; dragging them afterwards makes them instances: a random subset (or all) moved on
        r = np.random.default_rng(seed)
        xf = sc.obj_xform.copy()
        for ob in range(len(xf)):
            if r.random() < 0.6: xf[ob, 3::4] += (0.1 * float(np.abs(sc.pos).max()) * r.normal(size=3)).astype(np.float32)
        v.set_transforms(xf); o.set_transforms(xf)
        v.render(spp); o.render(spp)
        assert v.get_tlas() == o.get_tlas()
        assert np.array_equal(bits(v.read_hdr()), bits(o.read_hdr())), (seed, "after set_transforms", v.get_tlas())
        gs, cs = v.stats(), o.stats()
        for k in ("rays_nearest", "rays_any", "nodes_nearest", "tris_nearest", "nodes_any", "tris_any", "shaded_hits", "samples"):
            assert gs[k] == cs[k], (seed, k, gs[k], cs[k])


@pytest.mark.parametrize("seed", range(1, 49))
def test_random_call_sequences_keep_both_sides_in_step(view_cls, Oracle, seed):
    """State handling of the boundary under arbitrary call orders: renders, Redraws with speculative look-ahead, tile
    subsets, resets, and camera / material / light / environment / parameter changes WITHOUT a reset in between (the
    accumulation simply continues, pending look-ahead samples must be dropped).  After every step the images agree bit for bit."""
    import dataclasses
    r = np.random.default_rng(1000 + seed)
    sc = random_scene(seed + 200)
    sc = dataclasses.replace(sc, tri_object=None, obj_xform=None, params=dataclasses.replace(sc.params, max_depth=min(sc.params.max_depth, 6)))
    v = view_cls(0).load_scene(sc); o = Oracle().load_scene(sc)
    v.set_lookahead(int(r.choice([1, 3, 8])))
    if seed % 3 == 0: v.set_lookahead_auto(int(r.choice([2, 8, 16])))     # the ramping look-ahead takes the fixed one's place
    par, cam = sc.params, sc.camera
    done = 0                                             # whole-frame iterations since the last restart (what crh_render continues from)
    for step in range(14):
        op = int(r.integers(0, 10))
        if op <= 2:
            n = int(r.integers(1, 4)); v.render(n); o.render(n); done += n
        elif op == 3:
            for _ in range(int(r.integers(1, 4))):
                v.Redraw(); o.render(1); done += 1
        elif op == 4:
            tiles = np.arange(v.n_tiles(), dtype=np.uint32)
            sel = tiles[r.random(len(tiles)) < 0.5]
            if len(sel):
                v.render_tiles(sel, done, 1); o.render_tiles(sel, done, 1)
        elif op == 5:
            v.reset(); o.reset(); done = 0
        elif op == 6:
            eye = np.array(cam.eye) * (1 + 0.05 * r.normal()); cam = dataclasses.replace(cam, eye=tuple(eye), fovy_deg=float(r.choice([30.0, 60.0])))
            v.set_camera(cam); o.set_camera(cam)
        elif op == 7:
            mats = [random_bsdf(r, len(sc.textures)) for _ in sc.materials]
            v.set_materials(mats); o.set_materials(mats)
        elif op == 8:
            lights = [Light.directional(r.normal(size=3), 0.2, 3.0)] if r.random() < 0.7 else []
            env = scenes.procedural_sky(16, 8, int(seed + step)) if r.random() < 0.5 else None
            v.set_lights(lights); o.set_lights(lights); v.set_envmap(env); o.set_envmap(env)
        else:
            par = dataclasses.replace(par, width=int(r.integers(9, 60)), height=int(r.integers(9, 40)), max_depth=int(r.choice([1, 3, 6])),
                                      radiance_clamp=float(r.choice([0.0, 5.0])), tile_size=int(r.choice([8, 16, 32])))
            v.set_params(par); o.set_params(par); done = 0
        g, c = v.read_hdr(), o.read_hdr()
        assert np.array_equal(bits(g), bits(c)), (seed, step, op)
    assert np.array_equal(v.read_ldr(), o.read_ldr())


@pytest.mark.parametrize("seed", range(1, 25))
def test_random_sequences_two_level_adaptive_checkpoint(view_cls, Oracle, seed):
    """The rarer state transitions: object transforms moved between renders (top-level rebuild), adaptive screen sampling
    switched on and off, accumulator checkpoints restored into a fresh context -- still bit-identical to the oracle."""
    import dataclasses
    r = np.random.default_rng(5000 + seed)
    sc = None
    for s in range(seed * 7, seed * 7 + 200):                       # a two-level scene from the generator
        cand = random_scene(s)
        if cand.tri_object is not None:
            sc = cand; break
    sc = dataclasses.replace(sc, params=dataclasses.replace(sc.params, max_depth=min(sc.params.max_depth, 5), width=40, height=24, tile_size=8))
    v = view_cls(0).load_scene(sc); o = Oracle().load_scene(sc)
    v.set_lookahead(int(r.choice([1, 4])))
    xf = sc.obj_xform.copy()
    adaptive = False
    for step in range(10):
        op = int(r.integers(0, 6))
        if op <= 1:
            n = int(r.integers(1, 4)); v.render(n); o.render(n)
        elif op == 2:
            xf = xf.copy(); k = int(r.integers(0, len(xf))); xf[k, 3::4] += (0.05 * r.normal(size=3)).astype(np.float32)
            v.set_transforms(xf); o.set_transforms(xf)
        elif op == 3:
            adaptive = not adaptive
            v.set_adaptive(adaptive, 6); o.set_adaptive(adaptive, 6)
        elif op == 4 and not adaptive:
            acc, frames = v.save_accum()
            w = view_cls(0).load_scene(sc); w.set_transforms(xf); w.load_accum(acc, frames)      # the same build-time placement, then the same moves (a fresh build with xf would bake other trees)
            w.set_lookahead(int(r.choice([1, 4])))
            v = w                                                     # carry on in the restored context
        else:
            v.reset(); o.reset()
        assert np.array_equal(bits(v.read_hdr()), bits(o.read_hdr())), (seed, step, op)
    v.render(2); o.render(2)
    assert np.array_equal(bits(v.read_hdr()), bits(o.read_hdr()))


@pytest.mark.parametrize("seed", range(16))
def test_wide_batch_slot_layouts(view_cls, Oracle, seed):
    """The slot layout of wide batches (kernels.hip: 64 / G pixels x G samples per wavefront, G = the largest power of two dividing the
    batch's sample count, and the LDS-tiled accumulate): sample counts for every G, in one call, split in two, under look-ahead and
    through crh_render_tiles with a tile subset and a first-sample offset.  Image and counters against the oracle
    (tests/hunts/wide_batch_fuzz.py is the same hunt with more seeds)."""
    import dataclasses
    r = np.random.default_rng(seed)
    sc = random_scene(seed + 900)
    sc = dataclasses.replace(sc, params=dataclasses.replace(sc.params, width=int(r.integers(9, 50)), height=int(r.integers(9, 34)), max_depth=min(sc.params.max_depth, 4),
                                                            tile_size=int(r.choice([8, 16, 32]))))
    ns = [16, 24, 32, 40, 48, 64, 72, 96, 128, 130, 192, 8, 12, 56, 88, 160][seed]
    v = view_cls(0).load_scene(sc); o = Oracle().load_scene(sc)
    mode = seed % 4
    if mode == 0:
        v.enable_counters(True); v.reset(); v.render(ns); o.render(ns)
    elif mode == 1:
        k = int(r.integers(1, ns)); v.render(k); v.render(ns - k); o.render(ns)
    elif mode == 2:
        v.set_lookahead(int(r.choice([16, 32, 64]))); done = 0
        while done < ns:
            k = min(int(r.integers(1, 20)), ns - done); v.render(k); done += k
        o.render(ns)
    else:
        tiles = np.arange(v.n_tiles(), dtype=np.uint32); sel = tiles[r.random(len(tiles)) < 0.6]
        sel = sel if len(sel) else tiles[:1]
        first = int(r.integers(0, 50)); v.render_tiles(sel, first, ns); o.render_tiles(sel, first, ns)
    assert np.array_equal(bits(v.read_hdr()), bits(o.read_hdr())), (seed, ns, mode)
    if mode == 0:
        gs, cs = v.stats(), o.stats()
        for k in ("rays_nearest", "nodes_nearest", "tris_nearest", "rays_any", "samples"):
            assert gs[k] == cs[k], k


@pytest.mark.parametrize("seed", range(12))
def test_random_sequences_between_one_tree_and_object_trees(view_cls, Oracle, seed):
    """Scenes whose objects all sit at the identity are ONE tree; the first object that moves makes them object
    trees + top level, all-identity transforms flatten them again.  Random walks over that boundary -- renders, look-ahead, single
    moves, everything back in place, resets -- with the oracle in step after every call (images, counters, tree bytes)."""
    import dataclasses
    r = np.random.default_rng(7000 + seed)
    sc = None
    for s_ in range(seed * 5, seed * 5 + 200):
        cand = random_scene(s_)
        if cand.tri_object is not None:
            sc = cand; break
    nO = len(sc.obj_xform)
    ident = np.tile(np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], np.float32), (nO, 1))
    sc = dataclasses.replace(sc, obj_xform=ident.copy(), params=dataclasses.replace(sc.params, max_depth=min(sc.params.max_depth, 4), width=48, height=32, tile_size=8))
    v = view_cls(0).load_scene(sc); v.enable_counters(bool(seed % 2)); v.reset(); o = Oracle().load_scene(sc)
    assert v.get_tlas()["n_instances"] == 0 == o.get_tlas()["n_instances"]
    counted = bool(seed % 2)                                      # speculative look-ahead traces (and counts) frames the oracle has not rendered yet
    v.set_lookahead(1 if counted else int(r.choice([1, 4])))
    xf = ident.copy()
    for step in range(12):
        op = int(r.integers(0, 6))
        if op <= 1:
            n = int(r.integers(1, 4)); v.render(n); o.render(n)
        elif op == 2:
            xf = xf.copy(); k = int(r.integers(0, nO)); xf[k, 3::4] += (0.05 * r.normal(size=3)).astype(np.float32)
            v.set_transforms(xf); o.set_transforms(xf)
        elif op == 3:
            xf = ident.copy(); v.set_transforms(xf); o.set_transforms(xf)         # everything back in place
        elif op == 4:
            v.reset(); o.reset()
        elif not counted:
            v.set_lookahead(int(r.choice([1, 3, 8])))
        assert np.array_equal(bits(v.read_hdr()), bits(o.read_hdr())), (seed, step, op)
        assert v.get_tlas() == o.get_tlas() and np.array_equal(v.get_bvh()[0].view(np.uint32), o.get_bvh()[0].view(np.uint32)), (seed, step, op)
    if counted:
        gs, cs = v.stats(), o.stats()
        for k in ("rays_nearest", "nodes_nearest", "tris_nearest", "rays_any", "nodes_any", "shaded_hits"):
            assert gs[k] == cs[k], k


def _random_rigid(r, scale):
    a = r.random() * 6.28; ax = r.normal(size=3); ax /= np.linalg.norm(ax); s = float(r.choice([1.0, 1.0, 0.7, 1.4]))
    x, y, z = ax; c, sn = np.cos(a), np.sin(a)
    R = np.array([[c + x * x * (1 - c), x * y * (1 - c) - z * sn, x * z * (1 - c) + y * sn],
                  [y * x * (1 - c) + z * sn, c + y * y * (1 - c), y * z * (1 - c) - x * sn],
                  [z * x * (1 - c) - y * sn, z * y * (1 - c) + x * sn, c + z * z * (1 - c)]]) * s
    if r.random() < 0.4:
        R = np.eye(3)                                                # translation only (the instance keeps the world ray's reciprocals)
    return np.concatenate([R, (0.15 * scale * r.normal(size=3))[:, None]], 1).astype(np.float32).reshape(12)


@pytest.mark.parametrize("seed", range(16))
def test_random_walks_over_the_static_moved_split(view_cls, Oracle, seed):
    """Static / moved split under random call sequences: 2 .. 12 objects, some of them off the identity when the scene is built
    (instances for good), then single objects dragged away (translated, rotated, scaled), put back, everything put back, renders, resets -- at most four
    moved objects take the two-pass traversal, more the one-walk kernels, and the walk crosses that boundary.  After every call: image, top-level info,
    node and triangle-record bytes equal the oracle's; at the end all eight counters (odd seeds run the counting kernels)."""
    import dataclasses
    r = np.random.default_rng(9100 + seed)
    base = None
    for s_ in range(seed * 7, seed * 7 + 300):
        cand = random_scene(s_)
        if len(cand.tri) >= 60:
            base = cand; break
    n = len(base.tri); scale = float(np.abs(base.pos).max())
    nO = int(r.integers(2, 13))
    tri_obj = (np.arange(n) * nO // n).astype(np.int32) if r.random() < 0.5 else (np.arange(n) % nO).astype(np.int32)
    ident = np.tile(np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], np.float32), (nO, 1))
    xf0 = ident.copy()
    for k in range(nO):
        if r.random() < 0.25: xf0[k] = _random_rigid(r, scale)      # never part of the static tree
    sc = dataclasses.replace(base, tri_object=tri_obj, obj_xform=xf0.copy(),
                             params=dataclasses.replace(base.params, max_depth=min(base.params.max_depth, 4), width=48, height=32, tile_size=8))
    counted = bool(seed % 2)
    v = view_cls(0).load_scene(sc); v.enable_counters(counted); v.reset(); o = Oracle().load_scene(sc)
    xf = xf0.copy()
    seen_modes = set()
    for step in range(14):
        op = int(r.integers(0, 7))
        if op <= 1:
            k = int(r.integers(1, 3)); v.render(k); o.render(k)
        elif op == 2 or op == 3:
            xf = xf.copy(); xf[int(r.integers(0, nO))] = _random_rigid(r, scale); v.set_transforms(xf); o.set_transforms(xf)
        elif op == 4:
            xf = xf.copy(); xf[int(r.integers(0, nO))] = ident[0]; v.set_transforms(xf); o.set_transforms(xf)      # one object back in place
        elif op == 5:
            xf = ident.copy() if r.random() < 0.5 else xf0.copy(); v.set_transforms(xf); o.set_transforms(xf)
        else:
            v.reset(); o.reset()
        info = v.get_tlas()
        seen_modes.add(min(info["n_instances"], 5))
        assert info == o.get_tlas(), (seed, step, op)
        assert np.array_equal(bits(v.read_hdr()), bits(o.read_hdr())), (seed, step, op, info)
        gn, gt = v.get_bvh(); on, ot = o.get_bvh()
        assert np.array_equal(gn.view(np.uint32), on.view(np.uint32)) and np.array_equal(gt.view(np.uint32), ot.view(np.uint32)), (seed, step, op)
    v.render(2); o.render(2)
    assert np.array_equal(bits(v.read_hdr()), bits(o.read_hdr())), seed
    assert np.array_equal(v.read_ldr(), o.read_ldr())
    gs, cs = v.stats(), o.stats()
    for k in (("rays_nearest", "rays_any", "nodes_nearest", "tris_nearest", "nodes_any", "tris_any", "shaded_hits", "samples") if counted else ("rays_nearest", "rays_any", "shaded_hits", "samples")):
        assert gs[k] == cs[k], (seed, k, gs[k], cs[k])
