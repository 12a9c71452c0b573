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


def test_whole_harness_protocol_with_a_stand_in_program(tmp_path):
    """tools/cadrays_testing.py = testing/CADRays_Testing.py's command line and folder protocol (-i -c -f -d -o -m -u).  The program under
    test is a stand-in that writes what CADRays' test mode writes (main.cxx:193-228), so the protocol itself is checked without a GPU."""
    scripts, out, model = tmp_path / "scripts", tmp_path / "out", tmp_path / "model"
    for d in (scripts, out, model): d.mkdir()
    (scripts / "cornell.tcl").write_text("# a\n"); (scripts / "materials.TCL").write_text("# b\n"); (scripts / "notes.txt").write_text("x")
    (scripts / "Output_stale_5.txt").write_text("1")                 # left over from an earlier run: must be cleared
    prog = tmp_path / "fake_cadrays.py"
    prog.write_text(f"""#!{sys.executable}
import os, sys, zlib, struct
sys.path.insert(0, {os.path.join(ROOT, "tools")!r})
import numpy as np, compare_runs as cr
script, n = sys.argv[1], int(sys.argv[2])
name = os.path.splitext(os.path.basename(script))[0]
fps = float(os.environ.get("FAKE_FPS", "100")) * (2 if name == "materials" else 1)
img = np.full((8, 12, 3), len(name) * 9 % 256, np.uint8); img[2, 3, 0] = int(os.environ.get("FAKE_PIXEL", "7"))
open(f"Output_{{name}}_{{n}}.txt", "w").write("%g" % fps)
cr.write_png(f"Output_{{name}}_{{n}}.png", img)
""")
    os.chmod(prog, 0o755)
    tool = [sys.executable, os.path.join(ROOT, "tools", "cadrays_testing.py")]
    base = ["-i", str(scripts), "-c", str(prog), "-f", "5", "-o", str(out), "-m", str(model)]
    p = subprocess.run(tool + base, capture_output=True, text=True)       # first run: no template yet -> reported, not failed
    assert p.returncode == 0, p.stdout + p.stderr
    s = json.loads(p.stdout.strip().splitlines()[-1])
    assert [x["script"] for x in s["scripts"]] == ["cornell.tcl", "materials.tcl"] and all(x["fps_status"] == "no template" for x in s["scripts"])
    runs = [d for d in os.listdir(out)]
    assert len(runs) == 1 and not [f for f in os.listdir(scripts) if f.startswith("Output_")]            # outputs moved, stale ones cleared
    assert sorted(os.listdir(out / runs[0])) == ["Output_cornell_5.png", "Output_materials_5.png", "Result.html", "compare.json"]
    p = subprocess.run(tool + ["-u", "-o", str(out), "-m", str(model)], capture_output=True, text=True)   # accept it
    assert p.returncode == 0, p.stdout + p.stderr
    assert cr.read_template_rates(str(model)) == {"cornell.tcl": 100.0, "materials.tcl": 200.0} and sorted(os.listdir(model)) == ["Result.html", "cornell.png", "materials.png"]
    import time; time.sleep(1.1)                                      # the run folders are named by the second
    p = subprocess.run(tool + base, capture_output=True, text=True, env=dict(os.environ, FAKE_FPS="101"))
    s = json.loads(p.stdout.strip().splitlines()[-1])
    assert p.returncode == 0 and s["pass"] and all(x["fps_status"] == "same" and x["ldr_status"] == "identical" for x in s["scripts"])
    time.sleep(1.1)
    p = subprocess.run(tool + base + ["-d", "1"], capture_output=True, text=True, env=dict(os.environ, FAKE_FPS="97", FAKE_PIXEL="8"))
    s = json.loads(p.stdout.strip().splitlines()[-1])
    assert p.returncode == 1 and not s["pass"] and all(x["fps_status"] == "slower" and x["ldr_diff_pixels"] == 1 for x in s["scripts"])
    latest = sorted(os.listdir(out))[-1]
    assert os.path.isfile(out / latest / "Diff_cornell.png") and "background-color:red" in (out / latest / "Result.html").read_text()
    assert subprocess.run(tool + ["-i", str(tmp_path / "nope"), "-m", str(model)], capture_output=True).returncode == 2


@pytest.mark.gpu
def test_whole_harness_with_this_backend(hip_lib, tmp_path):
    """the same command line with the default program = this backend's script host: run, accept (-u), run again -> identical images"""
    import shutil
    scripts, out, model = tmp_path / "scripts", tmp_path / "out", tmp_path / "model"
    for d in (scripts, out, model): d.mkdir()
    src = open(os.path.join(ROOT, "tools", "material_preview.tcl")).read().replace("set frames_per_material 8000", "set frames_per_material 3")
    (scripts / "preview.tcl").write_text(src.replace("{brass bronze copper gold pewter plaster plastic silver steel stone shiny_plastic satin metalized neon_gnc chrome aluminium obsidian neon_phc jade charcoal water glass diamond transparent}", "{gold}"))
    tool = [sys.executable, os.path.join(ROOT, "tools", "cadrays_testing.py"), "-i", str(scripts), "-o", str(out), "-m", str(model), "-f", "5", "-d", "1000000"]
    p = subprocess.run(tool, capture_output=True, text=True)
    assert p.returncode == 0, p.stdout + p.stderr
    s = json.loads(p.stdout.strip().splitlines()[-1])
    assert not s["did_not_run"] and s["scripts"][0]["script"] == "preview.tcl" and s["scripts"][0]["fps"] > 0
    assert subprocess.run([sys.executable, os.path.join(ROOT, "tools", "cadrays_testing.py"), "-u", "-o", str(out), "-m", str(model)], capture_output=True).returncode == 0
    assert sorted(os.listdir(model)) == ["Result.html", "preview.pfm", "preview.png"]
    import time; time.sleep(1.1)
    p = subprocess.run(tool, capture_output=True, text=True)
    s = json.loads(p.stdout.strip().splitlines()[-1])
    assert p.returncode == 0 and s["pass"] and s["scripts"][0]["ldr_status"] == "identical" and s["scripts"][0]["hdr_rel_l2"] == 0.0
