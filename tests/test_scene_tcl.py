"""Scene wire format: the restricted Tcl evaluator, the reader of CADRays'
exported model.tcl + binary PLY, the writer that emits the exporter's layout, and -- when the reference tree is
mounted -- the reference's own demo scripts parsed into the same BSDF vectors the hand-restated fixtures hold."""
import dataclasses
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from cadrays_amd import scenes
from cadrays_amd.scene_tcl import MiniTcl, TclError, read_ply, read_scene, write_ply, write_scene

REF = "/root/reference/data/scripts"
needs_ref = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not mounted (GPU box)")


def abi_key(m):
    return tuple(np.round(np.frombuffer(bytes(m.to_abi()), np.float32), 6))


def test_mini_tcl_constructs():
    log = []
    t = MiniTcl({"emit": lambda a: log.append(" ".join(a))})
    t.eval('''
      # comment line
      set n 3
      for {set i 0} {$i < $n} {incr i} {
        for {set j 1} {$j <= 2} {incr j} {
          if {($i + $j) % 2 == 0} { emit even_[expr 12 * $i + $j] } else { emit odd_[expr $i * 10 - 90] }
        }
      }
      eval emit [lrepeat 3 x] tail
      emit "quoted $n" {braced $n}
    ''')
    assert log == ["odd_-90", "even_2", "even_13", "odd_-80", "odd_-70", "even_26", "x x x tail", "quoted 3 braced $n"]
    assert t.expr("7 / 2") == 3 and t.expr("7.0 / 2") == 3.5 and t.expr("(1 + 2) * 3 == 9") == 1
    with pytest.raises(TclError):
        t.eval("nosuchcommand 1 2")
    with pytest.raises(TclError):
        t.eval("emit $undefined")


def test_ply_round_trip(tmp_path):
    pos, nrm, tri = scenes.gen_scene(50, 3, 1)
    p = str(tmp_path / "m.ply")
    write_ply(p, pos, nrm, tri[:, :3])
    rp, rn, rf, uv = read_ply(p)
    assert np.array_equal(rp, pos) and np.array_equal(rn, nrm) and np.array_equal(rf, tri[:, :3]) and uv is None
    # ascii variant with a quad face and no normals
    (tmp_path / "a.ply").write_text("ply\nformat ascii 1.0\nelement vertex 4\nproperty float x\nproperty float y\nproperty float z\n"
                                    "element face 1\nproperty list uchar int vertex_indices\nend_header\n0 0 0\n1 0 0\n1 1 0\n0 1 0\n4 0 1 2 3\n")
    # without normals in the file the reference has assimp generate them (MeshImporter.cxx:80-87): one per FACE unless -gensmooth is given
    ap, an, af, _ = read_ply(str(tmp_path / "a.ply"))
    assert af.tolist() == [[0, 1, 2], [3, 4, 5]] and len(ap) == 6 and np.allclose(an, [0, 0, 1])
    sp, sn, sf, _ = read_ply(str(tmp_path / "a.ply"), smooth=True)
    assert sf.tolist() == [[0, 1, 2], [0, 2, 3]] and len(sp) == 4 and np.allclose(sn, [0, 0, 1])


def _ply_without_normals(path, with_uv):
    """a bent strip: smooth and per-face normals differ along the fold"""
    r = np.random.default_rng(4)
    xs = np.linspace(0, 1, 9)
    pos = np.array([[x, y, 0.35 * abs(x - 0.5) + 0.02 * r.standard_normal()] for y in (0.0, 0.5, 1.0) for x in xs], np.float32)
    faces = []
    for j in range(2):
        for i in range(8):
            a = j * 9 + i
            faces += [(a, a + 1, a + 10), (a, a + 10, a + 9)]
    uv = (pos[:, :2] * 2.0).astype(np.float32)
    with open(path, "wb") as f:
        f.write(("ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\n%selement face %d\n"
                 "property list uchar int vertex_index\nend_header\n" % (len(pos), "property float s\nproperty float t\n" if with_uv else "", len(faces))).encode())
        f.write(np.concatenate([pos] + ([uv] if with_uv else []), 1).astype("<f4").tobytes())
        rec = np.zeros(len(faces), np.dtype([("n", "u1"), ("i", "<i4", 3)])); rec["n"] = 3; rec["i"] = faces
        f.write(rec.tobytes())
    return pos, np.array(faces, np.int32), uv


def _normalless_model(tmp_path, flag, with_uv=True):
    d = tmp_path / ("m_" + (flag.strip("-") or "flat")); (d / "meshes").mkdir(parents=True)
    _ply_without_normals(str(d / "meshes" / "Strip.ply"), with_uv)
    model = d / "model.tcl"
    model.write_text("variable Root [file dirname [file normalize [info script]]]\nvclear\nvlight clear\n"
                     f"rtmeshread $Root/meshes/Strip.ply Strip {flag}\nvbsdf Strip -Kd 0.7 0.6 0.5 -Ks 0.2 0.2 0.2 -baseRoughness 0.3\n"
                     "vlight add directional direction -0.3 0.4 -1 smoothness 0.2 intensity 6\n"
                     "vcamera -persp -fovy 40\nvviewparams -eye 0.5 -1.6 1.4 -at 0.5 0.5 0.1 -up 0 0 1\nvrenderparams -ray -gi -rayDepth 4\n")
    return str(model)


def test_normal_less_ply_gets_flat_normals_unless_gensmooth(tmp_path):
    """MeshImporter.cxx:80-87: aiProcess_GenNormals (per-face normals, vertices split per triangle) without -gensmooth, aiProcess_GenSmoothNormals with it;
    the Python reader and the C++ reader of the saved-scene format hand over the same bytes either way (round-3 verdict item 7)."""
    from cadrays_amd import scene_io
    for flag in ("", "-gensmooth", "-gs"):
        model = _normalless_model(tmp_path, flag)
        sc, b = read_scene(model, 64, 48)
        assert not b.unsupported
        n_tri = 32
        if flag:
            assert len(sc.pos) == 27 and len(sc.tri) == n_tri                       # shared vertices kept, area-weighted normals
            assert len(np.unique(np.round(sc.nrm, 5), axis=0)) > 9
        else:
            assert len(sc.pos) == 3 * n_tri and len(sc.tri) == n_tri                # one vertex triple per face
            nr = sc.nrm.reshape(n_tri, 3, 3)
            assert np.array_equal(nr[:, 0], nr[:, 1]) and np.array_equal(nr[:, 0], nr[:, 2])
            e1 = sc.pos[1::3] - sc.pos[0::3]; e2 = sc.pos[2::3] - sc.pos[0::3]
            fn = np.cross(e1.astype(np.float64), e2.astype(np.float64)); fn /= np.linalg.norm(fn, axis=1, keepdims=True)
            assert np.allclose(nr[:, 0], fn, atol=1e-6)
            _, _, _, ruv = read_ply(os.path.join(os.path.dirname(model), "meshes", "Strip.ply"))
            assert ruv is not None and len(ruv) == 3 * n_tri                        # texture coordinates follow the split vertices
        a_path, b_path = tmp_path / f"py{flag}.crhscene", tmp_path / f"cpp{flag}.crhscene"
        scene_io.save_scene(sc, str(a_path))
        warn = _cpp_dump(model, b_path, "64x48")
        assert "not honoured" not in warn, warn
        assert a_path.read_bytes() == b_path.read_bytes(), f"readers disagree about a PLY without normals ({flag or 'no flag'})"


@pytest.mark.gpu
def test_normal_less_ply_renders_bit_exact_on_gpu(tmp_path, hip_lib, oracle_lib):
    from cadrays_amd.view import View
    imgs = {}
    for flag in ("", "-gensmooth"):
        sc, _ = read_scene(_normalless_model(tmp_path, flag, with_uv=False), 96, 72)
        v = View(0).load_scene(sc); v.render(4)
        o = oracle_lib.Oracle().load_scene(sc); o.render(4)
        assert np.array_equal(v.read_hdr().view(np.uint32), o.read_hdr().view(np.uint32)), flag
        imgs[flag] = v.read_hdr()
    assert not np.array_equal(imgs[""], imgs["-gensmooth"])                          # facets against a smooth fold


def test_export_format_round_trip(tmp_path):
    """write_scene emits what ImportExport::Export emits (model.tcl + meshes/*.ply); read_scene restores it."""
    sc = scenes.cornell_box(True, 64, 64)
    path = write_scene(sc, str(tmp_path))
    text = open(path).read()
    assert text.startswith("variable Root [file dirname [file normalize [info script]]]")
    assert "rtmeshread $Root/meshes/Mesh0.ply Mesh0 -group" in text and "vlight add positional position" in text
    back, b = read_scene(path, 64, 64)
    assert not b.unsupported and len(back.tri) == len(sc.tri) and len(back.materials) == len(sc.materials)
    assert [abi_key(m) for m in back.materials] == [abi_key(m) for m in sc.materials]
    # same triangle soup (the exporter groups triangles per object, so compare as sets of vertex triples)
    canon = lambda s: sorted(map(tuple, np.round(s.pos[s.tri[:, :3]].reshape(len(s.tri), 9), 6).tolist()))
    assert canon(back) == canon(sc)
    l0, l1 = sc.lights[0], back.lights[0]
    assert l1.is_point and np.allclose(l1.vec, l0.vec) and l1.smoothness == l0.smoothness and l1.intensity == l0.intensity
    assert np.allclose(back.camera.eye, sc.camera.eye) and np.allclose(np.array(back.camera.dir) / np.linalg.norm(back.camera.dir), [0, 1, 0])
    assert back.camera.fovy_deg == sc.camera.fovy_deg and back.params.max_depth == sc.params.max_depth


def test_vbsdf_vlocation_vlight_semantics(tmp_path):
    script = tmp_path / "s.tcl"
    script.write_text('''
      box b 0 0 0 1 2 3
      vdisplay b
      vsetmaterial b plastic
      vbsdf b -kd 1.0 0.8 0.2 -ks 0.3 -n
      vbsdf b -baseFresnel Schlick 0.58 0.42 0.2 -coatFresnel Dielectric 1.62 -Kc 1 1 1 -coatRoughness 0.05
      vlocation b -setLocation 1 0 0
      vlocation b -rotate 0 0 0 0 0 1 90
      vlight clear
      vlight add directional direction -0.25 -1 -1 sm 0.3 int 10
      vlight add ambient
      vlight add positional head 0 pos 0.5 0.5 0.85
      vlight change 2 sm 0.06
      vlight change 2 int 25.0
      rtlight 0 -color 1 0.5 0.25
      vcamera -persp -fovy 30
      vviewparams -eye 0 -5 1 -at 0 0 1 -up 0 0 1
      vrenderparams -ray -gi -rayDepth 7
    ''')
    sc, b = read_scene(str(script), 32, 32)
    m = sc.materials[0]
    np.testing.assert_allclose(m.Kd, np.array([1, 0.8, 0.2]) / 1.3, rtol=1e-6)          # -n == Normalize (MaterialEditor.cxx:311-329)
    np.testing.assert_allclose(m.Ks[:3], 0.3 / 1.3, rtol=1e-6)
    assert m.FresnelCoat.Serialize()[:2] == (-3.0, pytest.approx(1.62)) and m.Kc.tolist() == pytest.approx([1, 1, 1, 0.05])
    # local transformation = translation(1,0,0) * rotation(+90 deg about z): the 1x2x3 box turns in place, then moves
    assert np.allclose(sc.pos.min(0), [-1, 0, 0], atol=1e-6) and np.allclose(sc.pos.max(0), [1, 1, 3], atol=1e-6)
    assert len(sc.lights) == 2                                     # ambient is ignored by the path tracer
    assert sc.lights[0].color == (1.0, 0.5, 0.25) and sc.lights[0].smoothness == 0.3 and not sc.lights[0].is_point
    assert sc.lights[1].is_point and sc.lights[1].smoothness == 0.06 and sc.lights[1].intensity == 25.0
    assert sc.camera.fovy_deg == 30 and sc.params.max_depth == 7 and np.allclose(sc.camera.dir, [0, 5, 0])


@needs_ref
def test_reference_materials_script_matches_fixture():
    """data/scripts/Materials.tcl evaluated by the reader == cadrays_amd.scenes.materials_scene()."""
    sc, b = read_scene(os.path.join(REF, "Materials.tcl"), 256, 192)
    fx = scenes.materials_scene(256, 192)
    assert not b.unsupported and len(sc.tri) == len(fx.tri)
    assert {abi_key(m) for m in sc.materials} == {abi_key(m) for m in fx.materials}
    assert np.allclose(sc.camera.eye, fx.camera.eye) and sc.camera.fovy_deg == fx.camera.fovy_deg
    assert np.allclose(sc.lights[0].vec, fx.lights[0].vec) and sc.lights[0].intensity == 12 and sc.lights[0].smoothness == 0.3
    assert np.allclose(np.sort(sc.pos, 0), np.sort(fx.pos, 0), atol=1e-4)


@needs_ref
def test_reference_cornell_script():
    sc, b = read_scene(os.path.join(REF, "CornellBox.tcl"), 128, 128)
    assert not b.unsupported and sc.params.max_depth == 5
    l = sc.lights[0]
    assert l.is_point and l.vec == (0.5, 0.5, 0.85) and l.smoothness == 0.06 and l.intensity == 25.0
    kd = {tuple(float(x) for x in np.round(m.Kd.astype(np.float64), 4)) for m in sc.materials}
    assert (1.0, 0.3, 0.3) in kd and (0.3, 0.5, 1.0) in kd and (1.0, 1.0, 1.0) in kd
    # walls of the unit cube minus the front face: x in [0,1], y in [0,1], z in [0,1]
    assert np.allclose(sc.pos.min(0), 0, atol=1e-6) and np.allclose(sc.pos.max(0), 1, atol=1e-6)
    glass = [m for m in sc.materials if m.Kt.sum() > 0]
    assert len(glass) == 2 and {tuple(float(x) for x in np.round(g.Absorption.astype(np.float64), 3)) for g in glass} == {(0.8, 0.8, 1.0, 6.0), (0.8, 1.0, 0.8, 6.0)}


@pytest.mark.gpu
def test_parsed_scene_renders_bit_exact_on_gpu(tmp_path, hip_lib, oracle_lib):
    from cadrays_amd.view import View
    sc0 = scenes.materials_scene(96, 72, 16, 8)
    back, _ = read_scene(write_scene(sc0, str(tmp_path)), 96, 72)
    v = View(0).load_scene(back); v.render(3)
    o = oracle_lib.Oracle().load_scene(back); o.render(3)
    assert np.array_equal(v.read_hdr().view(np.uint32), o.read_hdr().view(np.uint32))


def _textured_export_scene():
    import dataclasses
    sc = scenes.cornell_box(True, 64, 64)
    uv = np.zeros((len(sc.pos), 2), np.float32)
    uv[:, 0] = sc.pos[:, 0] + 0.25 * sc.pos[:, 2]; uv[:, 1] = sc.pos[:, 1] - 0.5 * sc.pos[:, 2]
    r = np.random.default_rng(11)
    q = lambda a: (np.round(np.sqrt(a) * 255.0) / 255.0).astype(np.float32) ** 2     # values an 8-bit file can hold exactly
    rgb = q(r.random((8, 16, 3)).astype(np.float32))
    rgba = np.concatenate([q(r.random((4, 4, 3)).astype(np.float32)), (np.round(r.random((4, 4, 1)) * 255) / 255).astype(np.float32)], 2)
    mats = list(sc.materials)
    mats[2] = dataclasses.replace(mats[2], texture=0)
    mats[0] = dataclasses.replace(mats[0], texture=1)
    env = q(r.random((8, 16, 3)).astype(np.float32))
    return dataclasses.replace(sc, materials=mats, uv=uv, textures=[rgb, rgba], env=env)


def test_textures_and_uv_round_trip_through_the_export_layout(tmp_path):
    """rttexture lines + textures/*.png + s/t in the PLYs (ImportExport.cxx:235-264, :509; AisMesh.cxx:402-410)"""
    sc = _textured_export_scene()
    path = write_scene(sc, str(tmp_path))
    text = open(path).read()
    assert 'rttexture Mesh2 "$Root/textures/tex0.png"' in text and "vtextureenv on $Root/textures/env.png" in text
    assert os.path.exists(tmp_path / "textures" / "tex1.png")
    back, b = read_scene(path, 64, 64)
    assert not b.unsupported
    by_slot = {m.texture: m for m in back.materials if m.texture >= 0}
    assert len(by_slot) == 2 and len(back.textures) == 2
    # slots are renumbered in order of first use; compare by content
    want = {3: sc.textures[0], 4: sc.textures[1]}
    got = {t.shape[2]: t for t in back.textures}
    for ch in (3, 4):
        np.testing.assert_allclose(got[ch], want[ch], atol=1e-6)
    np.testing.assert_allclose(back.env, sc.env, atol=1e-6)
    # uv travel with their vertices
    key = lambda s: sorted(map(tuple, np.round(np.concatenate([s.pos[s.tri[:, :3]].reshape(len(s.tri), 9), s.uv[s.tri[:, :3]].reshape(len(s.tri), 6)], 1), 5).tolist()))
    assert key(back) == key(sc)


def test_rttexture_command_semantics(tmp_path):
    from cadrays_amd.scene_tcl import save_texture
    pos, nrm, tri = scenes.gen_scene(4, 2, 1)
    uv = np.random.default_rng(0).random((len(pos), 2)).astype(np.float32)
    write_ply(str(tmp_path / "m.ply"), pos, nrm, tri[:, :3], uv)
    write_ply(str(tmp_path / "n.ply"), pos, nrm, tri[:, :3])
    save_texture(str(tmp_path / "t.png"), np.full((2, 2, 3), 0.25, np.float32))
    s = tmp_path / "s.tcl"
    s.write_text(f'''
      rtmeshread {tmp_path}/m.ply A
      rtmeshread {tmp_path}/n.ply B
      box c 1 1 1
      vdisplay c
      rttexture A "{tmp_path}/t.png"
      rttexture A -scale 2 3
      rttexture B {tmp_path}/t.png
      rttexture c {tmp_path}/t.png
      vrenderparams -ray -gi -rayDepth 7 -iss
    ''')
    sc, b = read_scene(str(s), 32, 32)
    assert b.adaptive and sc.params.max_depth == 7
    assert sc.materials[0].texture == 0 and sc.materials[0].texture_scale == (1.0, 1.0)   # meshes keep their uv: -scale re-parametrises CAD shapes only
    assert sc.materials[1].texture == -1 and sc.materials[2].texture == -1
    assert len(b.unsupported) == 2 and all("no texture coordinates" in u for u in b.unsupported)
    np.testing.assert_allclose(sc.textures[0], 0.25, atol=2e-3)
    assert np.array_equal(sc.uv[:len(pos)], uv) and not sc.uv[len(pos):].any()
    # -off keeps the map but disables it; a missing file is an error like the reference's NoImageFile
    s.write_text(f'rtmeshread {tmp_path}/m.ply A\nrttexture A {tmp_path}/t.png\nrttexture A -off\n')
    sc2, _ = read_scene(str(s), 32, 32)
    assert sc2.materials[0].texture == -1 and sc2.uv is None
    s.write_text(f'rtmeshread {tmp_path}/m.ply A\nrttexture A {tmp_path}/missing.png\n')
    with pytest.raises(TclError):
        read_scene(str(s), 32, 32)


@pytest.mark.gpu
def test_script_host_runs_vfps_and_vdump_like_the_reference_test_mode(tmp_path, hip_lib, oracle_lib):
    """`CADRays <script.tcl> <nFrames>` (main.cxx:164-228): the script's own vfps / vdump are honoured live, then nFrames more
    Redraws end in Output_<name>_<n>.png / .txt; every image equals the oracle's LDR read-out of the same scene state."""
    import torch  # noqa: F401
    from PIL import Image
    from cadrays_amd.run_script import ScriptHost
    from cadrays_amd.view import View
    script = tmp_path / "demo.tcl"
    script.write_text('''
vinit name=View1 w=48 h=40
box floor -2 -2 -0.1 4 4 0.1
psphere ball 0.5
vdisplay floor ball
vsetlocation ball 0 0 0.5
vbsdf floor -kd 0.6
vlight del 1
vlight change 0 head 0 direction -0.25 -1 -1 sm 0.3 int 10
vcamera -persp
vviewparams -eye 3 -3 2 -at 0 0 0.4 -up 0 0 1
vrenderparams -ray -gi -rayDepth 4
foreach m {gold glass plaster} {
  vsetmaterial ball $m
  vfps 3
  vdump "D:/somewhere/$m.png"
}
''')
    host = ScriptHost(lambda: View(0), str(tmp_path))
    info = host.run(str(script), 2)
    assert info["frames"] == 3 * 3 + 2 and len(info["images"]) == 4 and not info["unsupported"]
    assert float((tmp_path / "Output_demo_2.txt").read_text()) > 0

    class OView(oracle_lib.Oracle):
        def Redraw(self):
            self.render(1)
    ref = tmp_path / "ref"; ref.mkdir()
    ScriptHost(lambda: OView(), str(ref)).run(str(script), 2)
    for name in ("gold.png", "glass.png", "plaster.png", "Output_demo_2.png"):
        a, b = np.asarray(Image.open(tmp_path / name)), np.asarray(Image.open(ref / name))
        assert a.shape == (40, 48, 3) and np.array_equal(a, b), name
    assert not np.array_equal(np.asarray(Image.open(tmp_path / "gold.png")), np.asarray(Image.open(tmp_path / "glass.png")))


def test_rtmeshread_obj_stl_and_up_axis(tmp_path):
    """OBJ and STL through rtmeshread, per-face vs -gensmooth normals (MeshImporter.cxx:73-91) and the -up mapping
    (MeshImporter.cxx:28-34)."""
    import struct
    from cadrays_amd.scene_tcl import MiniTcl, SceneBuilder, read_obj, read_stl
    # a unit square in the XY plane as two triangles + a quad OBJ with texture coordinates and negative indices
    (tmp_path / "q.obj").write_text("# quad\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvt 0 0\nvt 1 0\nvt 1 1\nvt 0 1\nf 1/1 2/2 3/3 4/4\n")
    pos, nrm, faces, uv = read_obj(str(tmp_path / "q.obj"))
    assert faces.shape == (2, 3) and len(pos) == 6 and np.allclose(nrm, [0, 0, 1]) and uv.shape == (6, 2)
    pos, nrm, faces, uv = read_obj(str(tmp_path / "q.obj"), smooth=True)
    assert len(pos) == 4 and np.allclose(nrm, [0, 0, 1])
    (tmp_path / "n.obj").write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nvn 0 0 -1\nf -3//1 -2//1 -1//1\n")
    pos, nrm, faces, uv = read_obj(str(tmp_path / "n.obj"))
    assert np.allclose(nrm, [0, 0, -1]) and uv is None and faces.tolist() == [[0, 1, 2]]
    tri = np.array([[[0, 0, 0], [1, 0, 0], [0, 1, 0]], [[1, 0, 0], [1, 1, 0], [0, 1, 0]]], np.float32)
    with open(tmp_path / "b.stl", "wb") as f:
        f.write(b"\0" * 80 + struct.pack("<I", 2))
        for t in tri:
            f.write(struct.pack("<12fH", 0, 0, 1, *t.reshape(-1), 0))
    (tmp_path / "a.stl").write_text("solid s\n" + "".join(
        "facet normal 0 0 1\nouter loop\n" + "".join("vertex %g %g %g\n" % tuple(v) for v in t) + "endloop\nendfacet\n" for t in tri) + "endsolid s\n")
    for name in ("b.stl", "a.stl"):
        pos, nrm, faces, _ = read_stl(str(tmp_path / name))
        assert pos.shape == (6, 3) and np.allclose(nrm, [0, 0, 1]) and np.allclose(pos.reshape(2, 3, 3), tri)
        pos, nrm, faces, _ = read_stl(str(tmp_path / name), smooth=True)
        assert pos.shape == (4, 3)
    b = SceneBuilder(str(tmp_path))
    t = MiniTcl(b.commands, {})
    t.eval(f"rtmeshread {tmp_path}/q.obj A\nrtmeshread {tmp_path}/b.stl B -gensmooth -up Y\nrtmeshread {tmp_path}/q.obj C -up -X")
    assert b.objs["A"].displayed and np.allclose(b.objs["A"].nrm, [0, 0, 1])
    assert np.allclose(b.objs["B"].nrm, [0, -1, 0]) and np.allclose(b.objs["B"].pos[:, 1], 0)          # (x, y, z) -> (x, -z, y)
    assert np.allclose(b.objs["C"].nrm, [1, 0, 0])                                                      # (x, y, z) -> (z, y, -x)
    with pytest.raises(TclError):
        t.eval(f"rtmeshread {tmp_path}/q.fbx D")


# ---- rtmeshread: materials of the imported mesh (AisMesh.cxx:228-346) and the plugin's options (ImportExportPlugin.cxx:132-354)
def _write_obj_with_mtl(d):
    """two quads and a box side with three materials, one with a diffuse map"""
    from PIL import Image
    tex = (np.arange(16 * 16 * 3).reshape(16, 16, 3) % 251).astype(np.uint8)
    Image.fromarray(tex, "RGB").save(d / "checker.png")
    (d / "room.mtl").write_text(
        "# materials\nnewmtl red_paint\nKa 0.1 0.1 0.1\nKd 0.9 0.2 0.1\nKs 0.5 0.5 0.5\nNs 98\n\n"
        "newmtl lamp\nKd 0.0 0.0 0.0\nKe 4 3 2\n\n"
        "newmtl tiles\nKd 1 1 1\nmap_Kd -s 1 1 1 checker.png\nmap_Ks spec.png\n")
    (d / "room.obj").write_text(
        "mtllib room.mtl\n"
        "v 0 0 0\nv 2 0 0\nv 2 2 0\nv 0 2 0\nv 0 0 1\nv 2 0 1\nv 2 2 1\nv 0 2 1\n"
        "vt 0 0\nvt 1 0\nvt 1 1\nvt 0 1\n"
        "o floor\nusemtl tiles\nf 1/1 2/2 3/3 4/4\n"
        "o ceiling\nusemtl lamp\nf 8/4 7/3 6/2 5/1\n"
        "o wall\nusemtl red_paint\nf 1/1 5/4 6/3 2/2\nusemtl tiles\nf 2/1 6/4 7/3 3/2\n"
        "o back\nf 4/1 3/2 7/3 8/4\n")
    return d / "room.obj"


def test_rtmeshread_imports_mtl_materials_like_aismesh(tmp_path):
    from cadrays_amd.materials import BSDF
    from cadrays_amd.scene_tcl import MiniTcl, SceneBuilder, mtl_to_bsdf, read_mtl, read_obj_meshes
    obj = _write_obj_with_mtl(tmp_path)
    mtl = read_mtl(str(tmp_path / "room.mtl"))
    assert set(mtl) == {"red_paint", "lamp", "tiles"} and mtl["tiles"]["map_Kd"].endswith("checker.png") and mtl["red_paint"]["Ns"] == 98
    # the conversion rule: Kd / Ks / Le copied, Ks.w = sqrt(2 / (Ns + 2)), then Normalize()
    b, tex = mtl_to_bsdf(mtl["red_paint"])
    s = np.float32(1.0) / np.float32(1.4)                                   # max channel of Kd + Ks = 0.9 + 0.5
    np.testing.assert_allclose(b.Kd[:3], np.float32([0.9, 0.2, 0.1]) * s, rtol=1e-6)
    np.testing.assert_allclose(b.Ks[:3], np.float32([0.5, 0.5, 0.5]) * s, rtol=1e-6)
    assert abs(b.Ks[3] - np.sqrt(2.0 / 100.0)) < 1e-7 and tex is None
    assert float(np.max(b.Kd[:3] + b.Ks[:3] + b.Kt[:3])) <= 1.0 + 1e-6
    b, tex = mtl_to_bsdf(mtl["lamp"])
    assert np.allclose(b.Le[:3], [4, 3, 2]) and np.allclose(b.Kd[:3], 0)
    b, tex = mtl_to_bsdf(mtl["tiles"])
    assert tex == str(tmp_path / "checker.png") and np.allclose(b.Kd[:3], 1)
    d, _ = mtl_to_bsdf(None)
    assert np.allclose(d.Kd[:3], BSDF.CreateDiffuse(0.8).Kd[:3])             # a mesh without a material: CreateDiffuse(0.8), AisMesh.cxx:246
    # one mesh per (object, material) run; -group merges by material
    meshes, _ = read_obj_meshes(str(obj))
    assert [(m["name"], m["material"]) for m in meshes] == [("floor", "tiles"), ("ceiling", "lamp"), ("wall", "red_paint"), ("wall", "tiles"), ("back", "tiles")]
    grouped, _ = read_obj_meshes(str(obj), group_by_material=True)
    assert [m["material"] for m in grouped] == ["tiles", "lamp", "red_paint"] and len(grouped[0]["faces"]) == 6

    b = SceneBuilder(str(tmp_path)); t = MiniTcl(b.commands, {})
    t.eval(f"rtmeshread {obj} room")
    assert b.groups["room"] == ["floor", "ceiling", "wall", "wall_1", "back"] and "room" not in b.objs
    assert b.objs["floor"].texture == str(tmp_path / "checker.png") and b.objs["floor"].tex_on and b.objs["ceiling"].texture is None
    assert np.allclose(b.objs["ceiling"].bsdf.Le[:3], [4, 3, 2]) and abs(b.objs["wall"].bsdf.Ks[3] - np.sqrt(0.02)) < 1e-7
    # commands addressed to the parent reach every sub-node
    t.eval("vlocation room -location 1 2 3\nverase room\nvdisplay room\nvbsdf wall_1 -kd 0.25")
    assert all(np.allclose(b.objs[n].t, [1, 2, 3]) and b.objs[n].displayed for n in b.groups["room"])
    sc = b.snapshot(32, 32)
    assert len(sc.materials) == 5 and len(sc.textures) == 1 and sc.uv is not None
    assert [m.texture for m in sc.materials] == [0, -1, -1, 0, 0] or [getattr(m, "texture", None) for m in sc.materials].count(0) == 3
    # the plugin's error behaviour
    with pytest.raises(TclError, match="already exists"):
        t.eval(f"rtmeshread {obj} room")
    t.eval(f"rtmeshread {obj} room -rename -group")
    assert len(b.groups["room_1"]) == 3
    with pytest.raises(TclError, match="usage"):
        t.eval(f"rtmeshread {obj} other -nosuchflag")
    with pytest.raises(TclError, match="usage"):
        t.eval(f"rtmeshread {obj} 9lives")
    t.eval(f"rtmeshread {obj} third -pretrans -genuv -pt -uv")             # accepted: no node hierarchy / no non-UV mapping in an OBJ


def test_rtmeshread_fixnorms_flips_inward_normals(tmp_path):
    from cadrays_amd.scene_tcl import MiniTcl, SceneBuilder, fix_infacing_normals
    m = scenes._Mesh(); m.box((1, 2, 3), 0, (0, 0, 0))
    pos, nrm, tri = m.arrays()
    p, n, f, flipped = fix_infacing_normals(pos, nrm, tri[:, :3])
    assert not flipped and np.array_equal(n, nrm)
    p, n, f, flipped = fix_infacing_normals(pos, -nrm, tri[:, :3])
    assert flipped and np.allclose(n, nrm) and np.array_equal(f, tri[:, :3][:, ::-1])
    quad = np.float32([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0]])           # flat: no inside, left alone
    p, n, f, flipped = fix_infacing_normals(quad, np.tile(np.float32([0, 0, -1]), (4, 1)), np.int32([[0, 1, 2], [0, 2, 3]]))
    assert not flipped
    lines = ["v %g %g %g" % tuple(x) for x in pos] + ["vn %g %g %g" % tuple(-x) for x in nrm]
    lines += ["f %d//%d %d//%d %d//%d" % (a + 1, a + 1, b + 1, b + 1, c + 1, c + 1) for a, b, c in tri[:, :3]]
    (tmp_path / "inward.obj").write_text("\n".join(lines) + "\n")
    b = SceneBuilder(str(tmp_path)); t = MiniTcl(b.commands, {})
    t.eval(f"rtmeshread {tmp_path}/inward.obj A\nrtmeshread {tmp_path}/inward.obj B -fixnorms")
    assert np.allclose(b.objs["A"].nrm, -nrm) and np.allclose(b.objs["B"].nrm, nrm)


@pytest.mark.gpu
def test_obj_with_mtl_renders_bit_exact_on_gpu(hip_lib, oracle_lib, tmp_path):
    from cadrays_amd.scene_tcl import MiniTcl, SceneBuilder
    from cadrays_amd.view import View
    obj = _write_obj_with_mtl(tmp_path)
    b = SceneBuilder(str(tmp_path)); t = MiniTcl(b.commands, {})
    t.eval(f"rtmeshread {obj} room\nvcamera -persp\nvviewparams -eye 0.25 0.3 0.45 -at 2 1.8 0.6 -up 0 0 1\nvrenderparams -ray -gi -rayDepth 5\nvlight del 0\nvlight del 1")
    sc = b.snapshot(64, 48)
    v = View(0).load_scene(sc); v.render(4)
    o = oracle_lib.Oracle().load_scene(sc); o.render(4)
    a, r = v.read_hdr(), o.read_hdr()
    assert np.array_equal(a.view(np.uint32), r.view(np.uint32)) and a.max() > 0.1          # lit by the emissive ceiling of the .mtl


# ---- the C++ reader of the saved-scene format (host/model_tcl.hpp; the reference host is C++ and re-imports by sourcing model.tcl,
# ImportSettingsEditor.cxx:378-380) must understand a model.tcl exactly like the Python reader does
def _cpp_dump(model, out, size="80x60"):
    import subprocess
    host = os.path.join(ROOT, "cadrays_amd", "host")
    exe = os.path.join(host, "model_tcl_dump")
    if not os.path.exists(exe) or max(os.path.getmtime(os.path.join(host, f)) for f in ("model_tcl.hpp", "jpeg_baseline.hpp", "model_tcl_dump.cpp")) > os.path.getmtime(exe):
        subprocess.check_call(["make", "-s", "-C", host, "model_tcl_dump"])
    p = subprocess.run([exe, str(model), str(out), size], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    return p.stderr


def _exported_scene(tmp_path):
    sc = scenes.materials_scene(80, 60, 16, 8)
    r = np.random.default_rng(5)
    uv = (r.random((len(sc.pos), 2)) * 3).astype(np.float32)
    tex = (r.random((8, 8, 3)) * 0.9 + 0.05).astype(np.float32)
    atex = np.concatenate([(r.random((4, 4, 3))).astype(np.float32), (r.random((4, 4, 1)) > 0.5).astype(np.float32)], 2)
    mats = [dataclasses.replace(m) for m in sc.materials]
    mats[0] = dataclasses.replace(mats[0], texture=0); mats[2] = dataclasses.replace(mats[2], texture=1)
    env = (r.random((16, 32, 3)) * 0.8).astype(np.float32)
    lights = list(sc.lights) + [scenes.Light.positional((0.5, -2.0, 3.0), smoothness=0.25, intensity=40.0, color=(1.0, 0.5, 0.25))]
    sc = dataclasses.replace(sc, uv=uv, textures=[tex, atex], materials=mats, env=env, lights=lights)
    from cadrays_amd.scene_tcl import write_scene
    return sc, write_scene(sc, str(tmp_path / "export"))


def test_cpp_reader_understands_model_tcl_like_the_python_reader(tmp_path):
    from cadrays_amd import scene_io
    from cadrays_amd.scene_tcl import read_scene
    sc, model = _exported_scene(tmp_path)
    with open(model, "a") as f:                                   # what the exporter adds for moved objects (ImportExport.cxx:276-305)
        f.write("vlocation Mesh1 -scale 1.5\nvlocation Mesh1 -location 0.25 -0.5 0.125\nvlocation Mesh3 -location 0 0 0.0625\n")
    py, b = read_scene(model, 80, 60)
    a_path, b_path = tmp_path / "py.crhscene", tmp_path / "cpp.crhscene"
    scene_io.save_scene(py, str(a_path))
    warn = _cpp_dump(model, b_path)
    assert "not honoured" not in warn and not b.unsupported
    A, B = a_path.read_bytes(), b_path.read_bytes()
    assert len(A) == len(B) and A == B, "the C++ reader and the Python reader disagree about model.tcl"
    # the straight-line subset is all the exporter writes; anything else is reported, not guessed
    loop = os.path.join(os.path.dirname(model), "loop.tcl")
    open(loop, "w").write(open(model).read() + "\nfor {set i 0} {$i < 3} {incr i} { vdisplay Mesh0 }\n")
    assert "not honoured: for" in _cpp_dump(loop, tmp_path / "x.crhscene")


def _cpp_image(path, tmp_path):
    import struct, subprocess
    host = os.path.join(ROOT, "cadrays_amd", "host"); exe = os.path.join(host, "model_tcl_dump")
    if not os.path.exists(exe) or max(os.path.getmtime(os.path.join(host, f)) for f in ("model_tcl.hpp", "jpeg_baseline.hpp", "model_tcl_dump.cpp")) > os.path.getmtime(exe):
        subprocess.check_call(["make", "-s", "-C", host, "model_tcl_dump"])
    out = str(tmp_path / "img.raw")
    p = subprocess.run([exe, "--image", str(path), out], capture_output=True, text=True)
    if p.returncode: return p.stderr.strip()
    d = open(out, "rb").read(); w, h, ch = struct.unpack("<3I", d[:12])
    return np.frombuffer(d[12:], np.uint8).reshape(h, w, ch)


def test_cpp_jpeg_reader_matches_pillow(tmp_path):
    """The environment map CADRays ships and loads by default is a baseline JPEG (data/maps/default.jpg, AppGui.cxx:963), so exported
    scenes reference one; host/jpeg_baseline.hpp must decode to the very bytes Pillow gives the Python reader: sequential and
    progressive, 4:4:4 / 4:2:2 / 4:2:0, sizes that are not MCU multiples, grey, restart intervals, optimised tables, three quality levels."""
    from PIL import Image
    r = np.random.default_rng(1)
    cases = 0
    for (w, h) in [(64, 48), (67, 53), (17, 9), (1, 1), (8, 8), (33, 16), (200, 131), (2, 37), (4, 16), (3, 3)]:      # the narrow ones: chroma planes of <= 2 samples are replicated, not filtered
        y, x = np.mgrid[0:h, 0:w]
        img = np.stack([127 + 120 * np.sin(x / 7.0 + y / 11.0), 127 + 120 * np.cos(x / 5.0), (x * 3 + y * 5) % 256], -1) + r.normal(0, 12, (h, w, 3))
        img = np.clip(img, 0, 255).astype(np.uint8)
        variants = [dict(quality=q, subsampling=sub, progressive=pr) for pr in (False, True) for sub in (0, 1, 2) for q in (30, 75, 95)]
        for pr in (False, True):
            variants += [dict(quality=80, subsampling=2, restart_marker_blocks=3, progressive=pr), dict(quality=80, optimize=True, progressive=pr),
                         dict(quality=80, grey=True, progressive=pr)]
        for kw in variants:
            src = img[..., 0] if kw.pop("grey", False) else img
            path = tmp_path / "t.jpg"; Image.fromarray(src).save(path, **kw)
            got = _cpp_image(path, tmp_path); want = np.asarray(Image.open(path).convert("RGB"))
            assert not isinstance(got, str), got
            assert got.shape == want.shape and np.array_equal(got, want), (w, h, kw)
            cases += 1
    assert cases == 240
    path = tmp_path / "c.jpg"; Image.fromarray(img).convert("CMYK").save(path)
    assert "3-component" in _cpp_image(path, tmp_path)                # refused with a message, not mis-decoded
    ref = "/root/reference/data/maps/default.jpg"
    if os.path.exists(ref):
        assert np.array_equal(_cpp_image(ref, tmp_path), np.asarray(Image.open(ref).convert("RGB")))


def test_cpp_png_writer_round_trips(tmp_path):
    """Output_<name>_<n>.png of the C++ host (the reference's dump format, main.cxx:193-228): what host/model_tcl.hpp writes must read
    back -- through Pillow and through its own reader -- as the pixels it was given."""
    import subprocess
    from PIL import Image
    _cpp_image(tmp_path / "missing.png", tmp_path)                  # builds the tool when needed
    exe = os.path.join(ROOT, "cadrays_amd", "host", "model_tcl_dump")
    r = np.random.default_rng(3)
    for shape in [(5, 7, 3), (64, 33, 4), (1, 1, 3), (300, 200, 3)]:
        a = (r.random(shape) * 255).astype(np.uint8)
        Image.fromarray(a).save(tmp_path / "in.png")
        subprocess.check_call([exe, "--to-png", str(tmp_path / "in.png"), str(tmp_path / "out.png")])
        assert np.array_equal(np.asarray(Image.open(tmp_path / "out.png")), a)
        assert np.array_equal(_cpp_image(tmp_path / "out.png", tmp_path), a)


def test_cpp_reader_loads_a_jpeg_environment_like_the_python_reader(tmp_path):
    from PIL import Image
    from cadrays_amd import scene_io
    from cadrays_amd.scene_tcl import read_scene
    sc, model = _exported_scene(tmp_path)
    r = np.random.default_rng(9)
    y, x = np.mgrid[0:32, 0:64]
    env = np.clip(np.stack([128 + 100 * np.sin(x / 9.0), 128 + 100 * np.cos(y / 5.0), 60 + 2 * x], -1) + r.normal(0, 6, (32, 64, 3)), 0, 255).astype(np.uint8)
    d = os.path.dirname(model)
    Image.fromarray(env).save(os.path.join(d, "textures", "sky.jpg"), quality=90, subsampling=2)
    txt = open(model).read()
    import re
    txt2, n = re.subn(r"vtextureenv on \S+", "vtextureenv on $Root/textures/sky.jpg", txt)
    assert n == 1
    open(model, "w").write(txt2)
    py, b = read_scene(model, 80, 60)
    assert py.env.shape == (32, 64, 3) and not b.unsupported
    a_path, b_path = tmp_path / "py.crhscene", tmp_path / "cpp.crhscene"
    scene_io.save_scene(py, str(a_path)); _cpp_dump(model, b_path)
    assert a_path.read_bytes() == b_path.read_bytes()


@pytest.mark.gpu
def test_cpp_driver_renders_model_tcl_like_the_oracle(hip_lib, oracle_lib, tmp_path):
    import subprocess
    from cadrays_amd.scene_tcl import read_scene
    sc, model = _exported_scene(tmp_path)
    exe = os.path.join(ROOT, "cadrays_amd", "host", "cadrays_headless")
    p = subprocess.run([exe, model, "3", "0", "1", "1", "80x60"], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    out = os.path.join(os.path.dirname(model), "Output_model_3.pfm")
    with open(out, "rb") as f:
        assert f.readline().strip() == b"PF"; w, h = (int(x) for x in f.readline().split()); f.readline()
        img = np.frombuffer(f.read(), "<f4").reshape(h, w, 3)[::-1]
    py, _ = read_scene(model, 80, 60)
    o = oracle_lib.Oracle().load_scene(py); o.render(3)
    assert np.array_equal(img.view(np.uint32), o.read_hdr().view(np.uint32))
