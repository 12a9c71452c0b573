"""tools/occt_pin: the kit a maintainer with OCCT uses to pin parity (round-2 verdict item 2).  CPU: the comparer on synthetic
folders.  GPU: the kit end to end against this backend itself, and the committed expected/ images reproduced bit for bit."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools", "occt_pin"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import pin_kit  # noqa: E402
import compare_runs as cr  # noqa: E402


def _fake_kit(tmp, frames=4):
    r = np.random.default_rng(1)
    base = {s: r.random((12, 16, 3)).astype(np.float32) for s in pin_kit.SCENES}
    for k, name in enumerate(pin_kit.SETTINGS):
        d = os.path.join(tmp, name); os.makedirs(d)
        for s in pin_kit.SCENES:
            img = base[s] * (1.0 + 0.05 * k)
            cr.write_pfm(os.path.join(d, f"Output_{s}_{frames}.pfm"), img)
            cr.write_png(os.path.join(d, f"Output_{s}_{frames}.png"), (np.clip(img, 0, 1) * 255).astype(np.uint8))
    json.dump({"frames": frames, "scenes": list(pin_kit.SCENES), "settings": pin_kit.SETTINGS}, open(os.path.join(tmp, "kit.json"), "w"))
    return base


def test_compare_ranks_the_setting_the_other_renderer_agrees_with(tmp_path):
    kit = str(tmp_path / "kit"); os.makedirs(kit)
    base = _fake_kit(kit)
    occt = str(tmp_path / "occt"); os.makedirs(occt)
    names = list(pin_kit.SETTINGS)
    for s in pin_kit.SCENES:                         # the "other renderer" agrees with the third setting, up to a little noise; LDR only for one scene
        img = base[s] * (1.0 + 0.05 * 2) + 1e-4
        cr.write_png(os.path.join(occt, f"Output_{s}_4.png"), (np.clip(img, 0, 1) * 255).astype(np.uint8))
        if s == pin_kit.SCENES[0]:
            cr.write_pfm(os.path.join(occt, f"Output_{s}_4.pfm"), img)
    rep = pin_kit.compare(kit, occt)
    assert rep["best"] == {s: names[2] for s in pin_kit.SCENES}
    rows = {(r["scene"], r["setting"]): r for r in rep["rows"]}
    assert rows[(pin_kit.SCENES[0], names[2])]["hdr_rel_l2"] < 1e-3 < rows[(pin_kit.SCENES[0], names[0])]["hdr_rel_l2"]
    assert rows[(pin_kit.SCENES[1], names[2])]["hdr_rel_l2"] is None and rows[(pin_kit.SCENES[1], names[2])]["ldr_mean_abs"] < 1.0
    # the ready-to-paste defaults (round-3 verdict item 5).  In this fake kit setting k is the base image x (1 + 0.05 k) and the other renderer sits at
    # k = 2: settings 1 .. 3 are closer to it than the default (k = 0) and are adopted, k = 4 is about as far as the default, the rest are farther
    rec = rep["recommendation"]
    assert list(rec["adopted"])[:3] == names[1:4] and not set(names[5:]) & set(rec["adopted"])
    for n in names[1:4]:
        for k, v in pin_kit.SETTINGS[n].items():
            assert rec["spec"][k] == v
    assert rec["c_initialiser"].startswith("#define CRH_SPEC_DEFAULTS {(uint32_t)sizeof(crh_spec), ") and rec["c_initialiser"].count(",") == 12
    import io, contextlib
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        pin_kit.print_report(rep)
    assert "ready to paste" in buf.getvalue() and rec["c_initialiser"] in buf.getvalue()
    # a folder with images of another size is reported, not crashed on
    cr.write_png(os.path.join(occt, f"Output_{pin_kit.SCENES[0]}_4.png"), np.zeros((3, 3, 3), np.uint8))
    assert any("error" in r for r in pin_kit.compare(kit, occt)["rows"])


def test_kit_settings_cover_every_runtime_switch():
    from cadrays_amd import abi
    flipped = set()
    for spec in pin_kit.SETTINGS.values():
        flipped |= set(spec)
        assert set(spec) <= set(abi.SPEC_DEFAULTS)
    assert flipped == set(abi.SPEC_DEFAULTS)


@pytest.mark.gpu
def test_kit_end_to_end_against_this_backend(hip_lib, tmp_path):
    rep = pin_kit.selfcheck(8, (48, 40), str(tmp_path))
    sc = rep["selfcheck"]
    assert sc["default_reproduces_itself"]
    # the reference's own two scenes cannot see every switch (no texture / environment / coat-less transmission in them): the kit's third
    # scene must, and the random-number switch must show everywhere
    assert len(sc["switches_told_apart"]["Switches"]) == len(pin_kit.SETTINGS) - 1, sc["switches_told_apart"]
    assert all("uniform_32bit" in told and "eps_rule" in told for told in sc["switches_told_apart"].values())
    assert sc["every_switch_observable"]
    assert os.path.exists(os.path.join(str(tmp_path), "scenes", "Switches", "Switches.tcl"))
    assert set(rep["noise_floor"]) == set(pin_kit.SCENES) and all(v["hdr_rel_l2"] > 0 for v in rep["noise_floor"].values())


@pytest.mark.gpu
def test_committed_expected_images_are_reproduced(hip_lib, tmp_path):
    exp = os.path.join(ROOT, "tools", "occt_pin", "expected")
    meta = json.load(open(os.path.join(exp, "kit.json")))
    pin_kit.make(str(tmp_path), meta["frames"], tuple(meta["size"]), None)
    for name in pin_kit.SETTINGS:
        for scene in pin_kit.SCENES:
            f = f"Output_{scene}_{meta['frames']}"
            a, b = cr.read_pfm(os.path.join(exp, name, f + ".pfm")), cr.read_pfm(os.path.join(str(tmp_path), name, f + ".pfm"))
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (name, scene)
            assert np.array_equal(cr.read_png(os.path.join(exp, name, f + ".png")), cr.read_png(os.path.join(str(tmp_path), name, f + ".png")))
