"""The reference's own renders as vectors: the 25 material icons OCCT's path tracer wrote for data/other/preview.tcl:10-64 (data/materials/*.png,
loaded by src/Launcher/main.cxx:120-132) against this project's render of the same scene (tools/material_preview.tcl).

tests/golden/icon_features.json holds what tests/golden/make_icon_features.py measured in the icons (tools/icon_features.py: horizon, ball silhouette,
tile edges, highlight, shadow side, the floor seen through the refracting balls, caustic, tile contrast); here the oracle (CPU) and the HIP path
(GPU, through the C ABI) render the scene at the recipe's 128 x 128, are scaled down like the icons, and go through the SAME extraction.

What the icons cannot pin: radiance.  The recipe's environment map (preview.tcl:56) is a file on its author's disk and OCCT's stock BSDFs are not in the
reference; the scene here is lit by a constant grey environment instead.  Everything compared below is independent of both.  What they DO pin is
written down per test, and collected in DESIGN.md section 2.
"""
import dataclasses
import json
import math
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import icon_features as F  # noqa: E402

from cadrays_amd.materials import BSDF  # noqa: E402
from cadrays_amd.scene_tcl import MiniTcl, SceneBuilder  # noqa: E402

GOLDEN = json.load(open(os.path.join(ROOT, "tests", "golden", "icon_features.json")))
ICONS = GOLDEN["icons"]
ENV_LEVEL = 0.45          # stand-in for the absent environment map: the level at which the shadowed floor is as deep as in the icons (0.62 of the lit floor)
SPP = 96


def preview_builder():
    """tools/material_preview.tcl up to its render loop: camera, floor, ball, light of data/other/preview.tcl:10-59"""
    src = open(os.path.join(ROOT, "tools", "material_preview.tcl")).read()
    b = SceneBuilder(os.path.join(ROOT, "tools"))
    MiniTcl(b.commands, {}).eval(src[:src.index("foreach name")])
    return b


def preview_scene(stock=None, bsdf=None, smoothness=None, size=128):
    b = preview_builder()
    if stock:
        b.cmd_vsetmaterial(["probe", stock])
    if bsdf is not None:
        b.objs["probe"].bsdf = bsdf
    sc = b.snapshot(size, size, "preview_" + (stock or "custom"))
    assert (sc.params.width, sc.params.height, sc.params.max_depth) == (128, 128, 10) or size != 128      # vinit w=128 h=128, -rayDepth 10
    assert not sc.params.env_as_background
    if smoothness is not None:
        sc = dataclasses.replace(sc, lights=[dataclasses.replace(sc.lights[0], smoothness=smoothness)])
    return dataclasses.replace(sc, env=np.full((8, 16, 3), ENV_LEVEL, np.float32))


class Renderer:
    """renders once per (backend, key) and keeps the 64 x 64 picture"""

    def __init__(self, make_backend):
        self.make, self.cache = make_backend, {}

    def picture(self, key, scene_fn, spp=SPP, **spec):
        k = (key, spp, tuple(sorted(spec.items())))
        if k not in self.cache:
            b = self.make().load_scene(scene_fn())
            if spec:
                b.set_spec(**spec)
            b.render(spp)
            self.cache[k] = F.box_down(b.read_ldr())
            b.close()
        return self.cache[k]

    def features(self, key, scene_fn, spp=SPP, **spec):
        return F.extract(self.picture(key, scene_fn, spp, **spec), key)


@pytest.fixture(scope="module")
def cpu(oracle_lib):
    return Renderer(lambda: oracle_lib.Oracle())


@pytest.fixture(scope="module")
def gpu(hip_lib):
    from cadrays_amd.view import View
    return Renderer(lambda: View(0))


# ------------------------------------------------------------------------------------------------ the committed vectors themselves
def test_golden_file_is_what_the_generator_writes():
    """in the build container the icons are readable: the committed features are those of the files (the GPU box has no /root/reference: skipped there)"""
    d = "/root/reference/data/materials"
    if not os.path.isdir(d):
        pytest.skip("no reference checkout here")
    from PIL import Image
    assert len(ICONS) == 25
    for icon, want in ICONS.items():
        got = F.extract(np.asarray(Image.open(os.path.join(d, icon + ".png")).convert("RGB")), icon)
        for k in ("horizon", "cap_circle", "cap_runs", "tile_ratio", "background_max"):
            assert got[k] == want[k], (icon, k)


def test_every_icon_shows_the_same_geometry():
    """25 renders of one scene: horizon, silhouette and tile edges agree among the icons to a tenth of a pixel or so -- the spread that bounds how
    finely anything can be pinned against them"""
    c = GOLDEN["consensus"]
    for icon, f in ICONS.items():
        assert f["background_max"] == 0.0, icon                            # the environment map lights the scene and is NOT shown behind it
        assert max(abs(a - b) for a, b in zip(f["horizon"], c["horizon"])) <= 0.02, icon
    assert all(e[2] <= 0.4 for kind in ("rows", "cols") for edges in c[kind].values() for e in edges)
    assert sum(len(v) for v in c["rows"].values()) == 12 and sum(len(v) for v in c["cols"].values()) >= 12
    assert 1.98 <= c["display_gamma"]["median"] <= 2.02 and c["display_gamma"]["max"] < 2.1 and c["display_gamma"]["n"] == 25


# ------------------------------------------------------------------------------------------------ the comparisons (shared by CPU and GPU)
def check_geometry(r):
    """Pins: default fovy 45 deg; `vviewparams -scale` has no effect once -eye / -at are given under -persp; the up vector is orthonormalised against the
    view direction; image row 0 is the top; a `box` / `ttranslate` / `psphere` / `vsetlocation` scene lands where OCCT puts it.  Bar (round-5 verdict):
    silhouette IoU >= 0.97, tile edges within 1 px, on every icon."""
    worst = {"horizon": 0.0, "edge": 0.0, "iou": 1.0, "circle": 0.0}
    for stock, icon in F.STOCK_TO_ICON:
        want, got = ICONS[icon], r.features(stock, lambda s=stock: preview_scene(s))
        assert got["background_max"] == 0.0, stock
        dh = max(abs(a - b) for a, b in zip(want["horizon"], got["horizon"]))
        assert dh <= 0.25, (stock, want["horizon"], got["horizon"])
        rows = min(want["cap_rows"], got["cap_rows"])
        i = F.iou(F.mask_from_runs(want["cap_runs"], (want["cap_rows"], F.N))[:rows], F.mask_from_runs(got["cap_runs"], (got["cap_rows"], F.N))[:rows])
        assert i >= 0.97, (stock, i)
        dc = max(abs(a - b) for a, b in zip(want["cap_circle"], got["cap_circle"]))
        assert dc <= 1.0, (stock, want["cap_circle"], got["cap_circle"])       # the sub-pixel rim estimate depends on how bright the rim is (chrome's is dark)
        if icon in ("plastered", "stone"):
            assert dc <= 0.25, (stock, want["cap_circle"], got["cap_circle"])  # matte balls: the same rim on both sides
        for kind in ("rows", "cols"):
            for key, edges in want[kind].items():
                d, missing = F.match_edges(edges, got[kind][key], tol=1.0)
                assert not missing, (stock, kind, key, missing, got[kind][key])
                worst["edge"] = max(worst["edge"], d)
        worst["horizon"], worst["iou"], worst["circle"] = max(worst["horizon"], dh), min(worst["iou"], i), max(worst["circle"], dc)
    # the composite icon (main.cxx:130): geometry like the rest
    got = r.features("plaster", lambda: preview_scene("plaster"))
    assert F.match_edges(ICONS["custom"]["rows"]["60"], got["rows"]["60"])[1] == []
    assert worst["edge"] <= 0.75 and worst["horizon"] <= 0.05, worst            # what it actually is: well inside the bar
    return worst


def check_display_gamma(r):
    """Pins crh_spec.h #15: a lit 0.85 tile over its 0.45 neighbour reads (0.85 / 0.45) ** (1 / gamma); the icons say gamma 2.00 (1.91 .. 2.04), this
    project's default reproduces it, the 2.2 of rounds 1 - 5 does not.  Also pins `vbsdf -kd <one number>` = a grey Lambert weight."""
    ref = [ICONS[i]["display_gamma"] for _, i in F.STOCK_TO_ICON]
    assert abs(float(np.median(ref)) - 2.0) <= 0.03
    ours = [r.features(s, lambda s=s: preview_scene(s))["display_gamma"] for s in ("plaster", "stone", "chrome", "charcoal", "jade")]
    assert abs(float(np.median(ours)) - 2.0) <= 0.08, ours
    old = [r.features(s + "@2.2", lambda s=s: preview_scene(s), display_gamma22=1)["display_gamma"] for s in ("plaster", "stone", "chrome", "charcoal", "jade")]
    assert abs(float(np.median(old)) - 2.2) <= 0.1 and float(np.median(old)) - max(ref) > 0.08, (old, max(ref))
    return float(np.median(ours)), float(np.median(old))


def check_light(r):
    """Pins: `vlight change 0 direction -0.25 -1 -1` is the direction the light TRAVELS in world space with `head 0` (highlight on the upper right of the
    ball within a pixel, shadow to the left); `sm 0.3` is the half-angle of the light's cone IN RADIANS (the saturated disc on a delta reflector covers
    42 - 44 icon pixels on diamond / glass; this renderer gives 42 at 0.3 rad, 20 at 0.2, 73 at 0.4)."""
    for stock in F.MIRROR_LIKE + ("diamond",):
        icon = dict(F.STOCK_TO_ICON)[stock]
        want, got = ICONS[icon]["highlight"], r.features(stock, lambda s=stock: preview_scene(s))["highlight"]
        assert want and got, stock
        assert math.hypot(want[0] - got[0], want[1] - got[1]) <= 1.0, (stock, want, got)
    for stock, icon in F.STOCK_TO_ICON:
        want, got = ICONS[icon]["shadow_left_over_right"], r.features(stock, lambda s=stock: preview_scene(s))["shadow_left_over_right"]
        if stock in F.REFRACTIVE + ("transparent",):
            assert want > 0.8 and got > 0.8, stock                         # light passes: no dark side
        elif stock != "neon_phc":                                          # (the emissive ball lights its own shadow)
            assert want < 0.72 and got < 0.75, (stock, want, got)
    area = {}
    mirror = BSDF.Metal(1.0, 0.0, 1.0)
    for a in (0.2, 0.3, 0.4):
        area[a] = r.features("mirror@%g" % a, lambda a=a: preview_scene(bsdf=mirror, smoothness=a), spp=128)["highlight"][2]
    icon_area = [ICONS[i]["highlight"][2] for i in ("diamond", "glass")]
    assert all(abs(area[0.3] - x) <= 5 for x in icon_area), (area, icon_area)
    assert all(abs(area[0.2] - x) >= 15 and abs(area[0.4] - x) >= 15 for x in icon_area), (area, icon_area)
    return area, icon_area


def check_refraction(r):
    """Pins: Snell refraction through a dielectric coat + specular transmission, inside / outside bookkeeping, and the stock indices -- the floor seen
    through the ball (upside down, mirrored) correlates with the icon only at OCCT's index for that material; a caustic lies under glass and water where
    the icon has it (light through two delta interfaces reaches the floor only as an implicit hit of the cone light), none under the opaque balls."""
    table = {}
    for stock, ior in (("water", 1.33), ("glass", 1.62), ("diamond", 2.42)):
        want = ICONS[dict(F.STOCK_TO_ICON)[stock]]["interior"]
        corr = {}
        for n in (1.33, 1.62, 2.42):
            got = r.features("%s@%g" % (stock, n), lambda n=n: preview_scene(bsdf=BSDF.CreateGlass(1.0, (1, 1, 1), 0.0, n)))["interior"]
            corr[n] = F.correlation(want, got)
        table[stock] = corr
        assert corr[ior] >= 0.78, (stock, corr)
        assert all(corr[ior] - c >= 0.3 for n, c in corr.items() if n != ior), (stock, corr)
        # ... and the stock stand-in (same index + the absorption tint) as good or better
        got = r.features(stock, lambda s=stock: preview_scene(s))["interior"]
        assert F.correlation(want, got) >= 0.78, stock
    for stock, icon in F.STOCK_TO_ICON:
        want, got = ICONS[icon]["caustic"], r.features(stock, lambda s=stock: preview_scene(s))["caustic"]
        if stock in ("water", "glass"):
            assert want and got and math.hypot(want[0] - got[0], want[1] - got[1]) <= 1.5, (stock, want, got)
        elif stock not in ("neon_phc",):
            assert want is None and got is None, (stock, want, got)
    # the "transparent" preset does not refract: the tile edges behind the ball stay where they are without it
    want, got = ICONS["transparent"]["interior"], r.features("transparent", lambda: preview_scene("transparent"))["interior"]
    assert F.correlation(want, got) >= 0.7
    return table


# ------------------------------------------------------------------------------------------------ CPU: the oracle
def test_oracle_geometry_matches_all_icons(cpu):
    check_geometry(cpu)


def test_oracle_display_gamma_matches_icons(cpu):
    check_display_gamma(cpu)


def test_oracle_light_conventions_match_icons(cpu):
    check_light(cpu)


def test_oracle_refraction_matches_icons(cpu):
    check_refraction(cpu)


# ------------------------------------------------------------------------------------------------ GPU: the HIP path through the C ABI
@pytest.mark.gpu
def test_hip_preview_matches_all_icons(gpu, cpu):
    worst = check_geometry(gpu)
    gam = check_display_gamma(gpu)
    area = check_light(gpu)
    table = check_refraction(gpu)
    print(json.dumps({"worst": worst, "gamma_default_vs_2.2": gam, "highlight_area_by_cone": area[0], "icon_highlight_area": area[1],
                      "interior_correlation_by_index": {k: {str(n): round(c, 3) for n, c in v.items()} for k, v in table.items()}}))
    # ... and the picture itself is the oracle's, byte for byte (the preview scene is one more parity case: 1 732 triangles, glass, cone light, environment)
    for stock in ("glass", "chrome", "neon_phc"):
        assert np.array_equal(gpu.picture(stock, lambda s=stock: preview_scene(s)), cpu.picture(stock, lambda s=stock: preview_scene(s))), stock


@pytest.mark.gpu
def test_hip_preview_script_through_run_script(hip_lib, tmp_path):
    """the recipe end to end on the product path: cadrays_amd.run_script evaluates tools/material_preview.tcl (vfps / vdump honoured live), writes one PNG
    per stock name; the dumps carry the icons' geometry"""
    import re
    from cadrays_amd.run_script import ScriptHost
    from cadrays_amd.view import View
    from PIL import Image
    src = open(os.path.join(ROOT, "tools", "material_preview.tcl")).read()
    src = re.sub(r"set frames_per_material \\d+", "set frames_per_material 64", src)
    script = tmp_path / "preview.tcl"
    script.write_text(src)
    rep = ScriptHost(lambda: View(0), str(tmp_path)).run(str(script), 0)
    assert len(rep["images"]) == 24 and not rep["unsupported"]
    for stock, icon in F.STOCK_TO_ICON:
        got = F.extract(F.box_down(np.asarray(Image.open(tmp_path / (stock + ".png")).convert("RGB"))), stock)
        want = ICONS[icon]
        assert max(abs(a - b) for a, b in zip(want["horizon"], got["horizon"])) <= 0.25, stock
        for key, edges in want["rows"].items():
            assert not F.match_edges(edges, got["rows"][key])[1], (stock, key)
