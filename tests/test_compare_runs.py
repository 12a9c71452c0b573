"""tools/compare_runs.py -- the comparer half of the reference's harness (testing/CADRays_Testing.py:42-51, 144-167, 226-230):
frame rate within +-max-diff %, per-pixel LDR diff image, HDR relative L2, -u promotion."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import compare_runs as cr  # noqa: E402


def make_run(d, name, n, fps, img, hdr=None):
    os.makedirs(d, exist_ok=True)
    base = os.path.join(d, f"Output_{name}_{n}")
    open(base + ".txt", "w").write("%g" % fps)
    cr.write_png(base + ".png", img)
    if hdr is not None:
        cr.write_pfm(base + ".pfm", hdr)


def test_promote_then_compare(tmp_path):
    r = np.random.default_rng(0)
    img = (r.random((20, 30, 3)) * 255).astype(np.uint8)
    hdr = r.random((20, 30, 3)).astype(np.float32) * 3
    run0, tpl = str(tmp_path / "run0"), str(tmp_path / "template")
    make_run(run0, "cornell", 100, 250.0, img, hdr)
    make_run(run0, "materials", 100, 120.0, img[::-1].copy())
    assert cr.main(["--template", tpl, "--run", run0, "-u"]) == 0
    assert cr.read_template_rates(tpl) == {"cornell.tcl": 250.0, "materials.tcl": 120.0}             # readable like the reference's Result.html
    assert os.path.isfile(os.path.join(tpl, "cornell.png")) and os.path.isfile(os.path.join(tpl, "cornell.pfm"))
    assert np.array_equal(cr.read_pfm(os.path.join(tpl, "cornell.pfm")), hdr)

    # identical images, frame rates inside +-2 %: pass
    run1 = str(tmp_path / "run1")
    make_run(run1, "cornell", 100, 253.0, img, hdr * np.float32(1 + 2e-5))
    make_run(run1, "materials", 100, 118.0, img[::-1].copy())
    s = cr.compare(tpl, run1)
    assert s["pass"] and [x["fps_status"] for x in s["scripts"]] == ["same", "same"]
    assert s["scripts"][0]["ldr_status"] == "identical" and s["scripts"][0]["hdr_status"] == "within tolerance"
    assert s["scripts"][1]["hdr_status"] == "no image"
    assert os.path.isfile(os.path.join(run1, "Result.html")) and os.path.isfile(os.path.join(run1, "Diff_cornell.png"))
    assert cr.read_template_rates(run1) == {"cornell.tcl": 253.0, "materials.tcl": 118.0}             # a run's report can itself be promoted / parsed

    # faster is reported, not failed; slower than the tolerance fails; one changed pixel fails and shows up in the diff image
    run2 = str(tmp_path / "run2")
    img2 = img.copy(); img2[3, 4, 1] ^= 1
    make_run(run2, "cornell", 100, 300.0, img2, hdr * np.float32(1.01))
    make_run(run2, "materials", 100, 110.0, img[::-1].copy())
    s = cr.compare(tpl, run2)
    c, m = s["scripts"]
    assert not s["pass"] and c["fps_status"] == "faster" and c["ldr_status"] == "differs" and c["ldr_diff_pixels"] == 1 and c["hdr_status"] == "differs"
    assert m["fps_status"] == "slower" and abs(m["fps_diff_pct"] - (110 / 120 - 1) * 100) < 1e-3 and not m["pass"]
    d = cr.read_png(os.path.join(run2, "Diff_cornell.png"))
    assert d[3, 4].max() == 255 and (d > 0).any(2).sum() == 1
    html = open(os.path.join(run2, "Result.html")).read()
    assert "background-color:green" in html and "background-color:red" in html
    # the CLI: exit status 1 on a regression, JSON on stdout
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "compare_runs.py"), "-m", tpl, "-o", run2, "-d", "2"], capture_output=True, text=True)
    assert p.returncode == 1 and json.loads(p.stdout)["pass"] is False
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "compare_runs.py"), "-m", tpl, "-o", run2, "-d", "30", "--hdr-tol", "0.1"], capture_output=True, text=True)
    assert p.returncode == 1                                                                            # the pixel still differs


def test_new_script_and_size_change(tmp_path):
    img = np.zeros((8, 8, 3), np.uint8)
    tpl, run = str(tmp_path / "t"), str(tmp_path / "r")
    make_run(run, "a", 10, 50.0, img)
    cr.promote(tpl, run)
    run2 = str(tmp_path / "r2")
    make_run(run2, "a", 10, 50.0, np.zeros((8, 9, 3), np.uint8))
    make_run(run2, "b", 10, 70.0, img)                                   # no template yet: reported, not failed
    s = cr.compare(tpl, run2)
    a, b = s["scripts"]
    assert a["ldr_status"] == "size differs" and not a["pass"]
    assert b["fps_status"] == "no template" and b["ldr_status"] == "no template" and b["pass"]
    assert not cr.compare(tpl, str(tmp_path))["pass"]                    # a folder without outputs is not a passing run


@pytest.mark.gpu
def test_rendering_the_preview_script_twice_passes(hip_lib, tmp_path):
    """the runner half (cadrays_amd.run_script) feeding the comparer half: same script, same frames -> identical images"""
    from cadrays_amd.run_script import ScriptHost
    from cadrays_amd.view import View
    script = tmp_path / "preview.tcl"
    src = open(os.path.join(ROOT, "tools", "material_preview.tcl")).read().replace("set frames_per_material 8000", "set frames_per_material 4")
    script.write_text(src.replace("{brass bronze copper gold pewter plaster plastic silver steel stone shiny_plastic satin metalized neon_gnc chrome aluminium obsidian neon_phc jade charcoal water glass diamond transparent}", "{gold glass}"))
    runs = []
    for k in range(2):
        d = tmp_path / f"run{k}"; d.mkdir()
        info = ScriptHost(lambda: View(0), str(d), hdr=True).run(str(script), 6)
        assert info["frames"] == 2 * 4 + 6
        runs.append(str(d))
    tpl = str(tmp_path / "template")
    cr.promote(tpl, runs[0])
    s = cr.compare(tpl, runs[1], max_diff=1e9)                           # timing of a 14-frame run is noise; the images are the point
    assert s["pass"] and s["scripts"][0]["ldr_status"] == "identical" and s["scripts"][0]["hdr_rel_l2"] == 0.0
