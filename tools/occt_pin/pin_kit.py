#!/usr/bin/env python3
"""OCCT pinning kit: what somebody WITH an Open CASCADE build needs to turn "parity unpinned" into a number.

The arithmetic behind CADRays' `V3d_View::Redraw()` (src/Launcher/AppViewer.cxx:1047) lives in OCCT's TKOpenGl + GLSL shaders, which
are not in the reference repository and not in this image.  This backend's spec therefore contains choices that could not be
checked against the real renderer; every one of them is a switch (include/crh_spec.h).  The kit renders the reference's OWN two
parameterised scenes -- data/scripts/CornellBox.tcl and data/scripts/Materials.tcl, the only concrete scene + BSDF + light vectors the
reference holds for this path -- with this backend under the default spec and with each switch flipped, in the folder layout of the
reference's test mode (`CADRays <script.tcl> <nFrames>` writes Output_<name>_<n>.png / .txt, src/Launcher/main.cxx:199-221), and
compares a folder of OCCT outputs against all of them.

  make      python tools/occt_pin/pin_kit.py make --out DIR [--frames 64] [--size 128x128] [--cadrays-root /path/to/CADRays]
              DIR/<setting>/Output_<scene>_<frames>.png  (BufferDump RGB)  .pfm (linear HDR accumulator)  .txt (frames / s)
              DIR/kit.json  (settings, scenes, sizes, library, GPU)
            With --cadrays-root the scenes are the reference's scripts themselves (evaluated by cadrays_amd.scene_tcl); without it the
            hand-restated fixtures of cadrays_amd/scenes.py (same BSDF / light / camera vectors, cited line by line there).
  compare   python tools/occt_pin/pin_kit.py compare --kit DIR --occt OCCT_DIR
              OCCT_DIR holds what the real application wrote: Output_CornellBox_<n>.png / Output_Materials_<n>.png from
              `CADRays data/scripts/CornellBox.tcl <n>` (same window size, `vrenderparams -ray -gi -rayDepth` as in the script), and --
              if saved from the GUI as .hdr/.exr and converted -- Output_<scene>_<n>.pfm.  Prints, per scene and per switch setting, the
              LDR mean absolute difference / differing-pixel fraction and the HDR relative L2, ranks the settings, and says whether the
              best one is inside the noise floor (estimated from two seeds of this backend).
  selfcheck python tools/occt_pin/pin_kit.py selfcheck [--frames 16 --size 64x64]
              the whole flow against this backend itself: default vs default must be identical, every flipped switch must be told
              apart from it; prints the distance of each switch from the default and the seed-to-seed noise floor at that frame count.

Images of two different renderers never agree bit for bit (different random streams): the comparison is statistical, so use enough
frames that the noise floor (`selfcheck` prints it) is below the effect of the switch under test -- 4096 frames at 256 x 256 take
well under a minute on one MI355X.
"""
import argparse
import dataclasses
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

SETTINGS = {
    "default": {},
    "uniform_32bit": dict(uniform_32bit=1),
    "texel_gamma2": dict(texel_gamma2=1),
    "mis_single_lobe": dict(mis_single_lobe=1),
    "eps_rule": dict(eps_rule=1),
    "eta_no_dielectric_1.5": dict(eta_no_dielectric=1.5),
    # round 4: crh_spec.h #9 - #14
    "rr_start_bounce_1": dict(rr_start_bounce=1),
    "rr_survival_cap_0.25": dict(rr_survival_cap=0.25),
    "min_contribution_0.25": dict(min_contribution=0.25),
    "min_throughput_0.125": dict(min_throughput=0.125),
    "raygen_corners": dict(raygen_bilinear=1),
    "raygen_unit_corners": dict(raygen_bilinear=2),
    "env_orientation": dict(env_orientation=1),
    # round 6: crh_spec.h #15 (display only: the .pfm of this setting equals the default's, the .png does not)
    "display_gamma22": dict(display_gamma22=1),
}
SCENES = ("CornellBox", "Materials", "Switches")   # the reference's script names data/scripts/<name>.tcl, + this project's scene that exercises
                                                   # the switches those two cannot (no texture / environment / coat-less transmission in them);
                                                   # `make` exports it as DIR/scenes/Switches/Switches.tcl in the application's own format


def load_scene(name, w, h, cadrays_root):
    from cadrays_amd import scenes
    if name == "Switches":
        return scenes.spec_switch_scene(w, h)
    if cadrays_root:
        from cadrays_amd.scene_tcl import read_scene
        return read_scene(os.path.join(cadrays_root, "data", "scripts", name + ".tcl"), w, h)
    return scenes.cornell_box(True, w, h) if name == "CornellBox" else scenes.materials_scene(w, h, 32, 16)


def render_setting(view_cls, sc, frames, spec, seed=None):
    if seed is not None:
        sc = dataclasses.replace(sc, params=dataclasses.replace(sc.params, seed=seed))
    if spec.get("texel_gamma2"):
        # the switch moves the squaring from the file reader (before filtering) into the lookup (after filtering): hand over the RAW
        # 8-bit image values, as the real renderer's texture units hold them
        un = lambda t: None if t is None else np.concatenate([np.sqrt(t[..., :3]), t[..., 3:]], -1).astype(np.float32)
        sc = dataclasses.replace(sc, env=un(sc.env), textures=[un(t) for t in (sc.textures or [])])
    v = view_cls(0).load_scene(sc)
    v.set_spec(**spec)
    v.set_lookahead(min(256, max(1, (4 << 20) // (sc.params.width * sc.params.height))))
    t = time.perf_counter()
    for _ in range(frames):
        v.Redraw()                                         # one Redraw() per frame, like the reference's loop (AppViewer.cxx:1045-1047)
    v.sync()
    dt = time.perf_counter() - t
    hdr, ldr = v.read_hdr(), v.read_ldr()
    v.close()
    return hdr, ldr, frames / max(dt, 1e-9)


def make(out, frames, size, cadrays_root, settings=SETTINGS, seeds=(None,), write_meta=True, view_cls=None):
    """view_cls: the backend class (default the HIP path's View; tests/golden/make_occt_pin_expected.py passes the CPU checker, whose images are the same bits)"""
    import compare_runs as cr
    if view_cls is None:
        import torch  # noqa: F401  (runtime ordering: torch's HIP runtime first)
        from cadrays_amd.view import View
    else:
        View = view_cls
    w, h = size
    os.makedirs(out, exist_ok=True)
    meta = {"frames": frames, "size": [w, h], "scenes": list(SCENES), "settings": settings,
            "scene_source": "reference scripts under " + cadrays_root if cadrays_root else "cadrays_amd/scenes.py fixtures restating data/scripts/{CornellBox,Materials}.tcl",
            "layout": "SETTING/Output_<scene>_<frames>.png|.pfm|.txt (reference test mode, main.cxx:199-221)"}
    for sname, spec in settings.items():
        for seed in seeds:
            d = os.path.join(out, sname if seed is None else f"{sname}@seed{seed}")
            os.makedirs(d, exist_ok=True)
            for scene in SCENES:
                sc = load_scene(scene, w, h, cadrays_root)
                hdr, ldr, fps = render_setting(View, sc, frames, spec, seed)
                base = os.path.join(d, f"Output_{scene}_{frames}")
                cr.write_png(base + ".png", ldr); cr.write_pfm(base + ".pfm", hdr)
                open(base + ".txt", "w").write("%g" % fps)
    if write_meta:
        from cadrays_amd.scene_tcl import write_scene
        d = os.path.join(out, "scenes", "Switches")
        path = write_scene(load_scene("Switches", w, h, None), d)          # ImportExport::Export's layout: model.tcl + meshes/ + textures/
        os.replace(path, os.path.join(d, "Switches.tcl"))                   # test mode names its outputs after the script: Output_Switches_<n>.png
        json.dump(meta, open(os.path.join(out, "kit.json"), "w"), indent=1)
    return meta


def distance(a_dir, b_dir, scene, frames):
    """LDR and HDR distances between two output folders for one scene (None where a file is missing)"""
    import compare_runs as cr
    pa, pb = (os.path.join(d, f"Output_{scene}_{frames}") for d in (a_dir, b_dir))
    out = {"ldr_mean_abs": None, "ldr_diff_fraction": None, "hdr_rel_l2": None}
    if os.path.exists(pa + ".png") and os.path.exists(pb + ".png"):
        x, y = cr.read_png(pa + ".png").astype(np.int32), cr.read_png(pb + ".png").astype(np.int32)
        if x.shape == y.shape:
            out["ldr_mean_abs"] = float(np.abs(x - y).mean()); out["ldr_diff_fraction"] = float((np.abs(x - y).max(axis=2) > 0).mean())
        else:
            out["error"] = f"image sizes differ: {x.shape[1]}x{x.shape[0]} vs {y.shape[1]}x{y.shape[0]}"
    if os.path.exists(pa + ".pfm") and os.path.exists(pb + ".pfm"):
        x, y = cr.read_pfm(pa + ".pfm").astype(np.float64), cr.read_pfm(pb + ".pfm").astype(np.float64)
        if x.shape == y.shape:
            out["hdr_rel_l2"] = float(np.linalg.norm(x - y) / max(np.linalg.norm(y), 1e-300))
    return out


def compare(kit, occt):
    meta = json.load(open(os.path.join(kit, "kit.json")))
    frames = meta["frames"]
    rows = []
    for scene in meta["scenes"]:
        for sname in meta["settings"]:
            d = distance(occt, os.path.join(kit, sname), scene, frames)
            rows.append(dict(scene=scene, setting=sname, **d))
    key = lambda r: (r["hdr_rel_l2"] if r["hdr_rel_l2"] is not None else 1e9, r["ldr_mean_abs"] if r["ldr_mean_abs"] is not None else 1e9)
    report = {"frames": frames, "rows": rows, "best": {}}
    for scene in meta["scenes"]:
        sr = sorted([r for r in rows if r["scene"] == scene], key=key)
        if sr and (sr[0]["hdr_rel_l2"] is not None or sr[0]["ldr_mean_abs"] is not None):
            report["best"][scene] = sr[0]["setting"]
    noise = os.path.join(kit, "default@seed2")
    if os.path.isdir(noise):
        report["noise_floor"] = {s: distance(noise, os.path.join(kit, "default"), s, frames) for s in meta["scenes"]}
    report["recommendation"] = recommend(report, meta["settings"])
    return report


SPEC_FIELD_ORDER = ("uniform_32bit", "texel_gamma2", "mis_single_lobe", "eps_rule", "eta_no_dielectric", "rr_start_bounce", "rr_survival_cap",
                    "min_contribution", "min_throughput", "raygen_bilinear", "env_orientation", "display_gamma22")      # include/crh_spec.h, after `size`


def recommend(rep, meta_settings):
    """Which switches the other renderer agrees with: a flipped setting is ADOPTED when, in at least one scene, it is closer to the OCCT images than the
    default by more than the seed-to-seed noise floor (HDR rel L2 where there is a .pfm, LDR mean |d| otherwise) and in no scene farther by more than
    that.  Returns the spec as a dict and as a ready-to-paste initialiser for include/crh_spec.h."""
    from cadrays_amd import abi
    rows = {(r["scene"], r["setting"]): r for r in rep["rows"]}
    noise = rep.get("noise_floor") or {}
    spec = dict(abi.SPEC_DEFAULTS); adopted = {}; undecided = []
    for sname, flips in meta_settings.items():
        if sname == "default" or not flips:
            continue
        better = worse = 0
        for scene in {r["scene"] for r in rep["rows"]}:
            a, b = rows.get((scene, sname)), rows.get((scene, "default"))
            if not a or not b:
                continue
            for key in ("hdr_rel_l2", "ldr_mean_abs"):
                if a.get(key) is None or b.get(key) is None:
                    continue
                floor = (noise.get(scene) or {}).get(key) or 0.0
                if a[key] < b[key] - floor: better += 1
                elif a[key] > b[key] + floor: worse += 1
                break
        if better and not worse:
            spec.update(flips); adopted[sname] = flips
        elif not better and not worse:
            undecided.append(sname)
    f = lambda v: ("%gf" % v if "." in "%g" % v or "e" in "%g" % v else "%g.0f" % v) if isinstance(v, float) else str(int(v))
    init = "{(uint32_t)sizeof(crh_spec), " + ", ".join(f(float(spec[k])) if isinstance(abi.SPEC_DEFAULTS[k], float) else f(spec[k]) for k in SPEC_FIELD_ORDER) + "}"
    return {"adopted": adopted, "undecided_inside_noise": undecided, "spec": spec, "c_initialiser": "#define CRH_SPEC_DEFAULTS " + init}


def print_report(rep):
    print(f"{'scene':12s} {'setting':24s} {'LDR mean|d|':>12s} {'LDR px diff':>12s} {'HDR rel L2':>12s}")
    f = lambda v, p: "-" if v is None else p % v
    for r in rep["rows"]:
        print(f"{r['scene']:12s} {r['setting']:24s} {f(r['ldr_mean_abs'], '%.4f'):>12s} {f(r['ldr_diff_fraction'], '%.4f'):>12s} {f(r['hdr_rel_l2'], '%.3e'):>12s}")
    for s, n in (rep.get("noise_floor") or {}).items():
        print(f"{s:12s} {'(noise: 2 seeds, default)':24s} {f(n['ldr_mean_abs'], '%.4f'):>12s} {f(n['ldr_diff_fraction'], '%.4f'):>12s} {f(n['hdr_rel_l2'], '%.3e'):>12s}")
    print("closest setting per scene:", json.dumps(rep["best"]))
    if rep.get("recommendation"):
        r = rep["recommendation"]
        print("switches the other renderer agrees with (closer than the default by more than the noise floor):", json.dumps(r["adopted"]))
        print("inside the noise floor (render more frames to decide):", ", ".join(r["undecided_inside_noise"]) or "-")
        print("ready to paste into include/crh_spec.h (then regenerate the goldens):")
        print("  " + r["c_initialiser"])


def selfcheck(frames, size, out=None):
    import tempfile
    tmp = out or tempfile.mkdtemp(prefix="occt_pin_")
    make(tmp, frames, size, None)
    make(tmp, frames, size, None, settings={"default": {}}, seeds=(2,), write_meta=False)       # a second seed of the default: the noise floor
    # stand-in for the OCCT run: this backend itself, default spec, rendered again
    again = os.path.join(tmp, "_again")
    make(again, frames, size, None, settings={"default": {}})
    rep = compare(tmp, os.path.join(again, "default"))
    print_report(rep)
    ok = all(r["ldr_diff_fraction"] == 0.0 and r["hdr_rel_l2"] == 0.0 for r in rep["rows"] if r["setting"] == "default")
    told_apart = {s: [r["setting"] for r in rep["rows"] if r["scene"] == s and r["setting"] != "default" and ((r["hdr_rel_l2"] or 0) > 0 or (r["ldr_diff_fraction"] or 0) > 0)] for s in SCENES}
    covered = sorted(set(x for v in told_apart.values() for x in v))
    rep["selfcheck"] = {"default_reproduces_itself": ok, "switches_told_apart": told_apart, "every_switch_observable": len(covered) == len(SETTINGS) - 1, "dir": tmp}
    print(json.dumps(rep["selfcheck"]))
    return rep


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    sub = ap.add_subparsers(dest="cmd", required=True)
    size = lambda s: tuple(int(x) for x in s.lower().split("x"))
    m = sub.add_parser("make"); m.add_argument("--out", required=True); m.add_argument("--frames", type=int, default=64)
    m.add_argument("--size", type=size, default=(128, 128)); m.add_argument("--cadrays-root", default="")
    m.add_argument("--noise-seed", action="store_true", help="also render the default spec with seed 2 (noise floor for `compare`)")
    c = sub.add_parser("compare"); c.add_argument("--kit", required=True); c.add_argument("--occt", required=True)
    s = sub.add_parser("selfcheck"); s.add_argument("--frames", type=int, default=16); s.add_argument("--size", type=size, default=(64, 64)); s.add_argument("--out", default="")
    a = ap.parse_args(argv)
    if a.cmd == "make":
        print(json.dumps(make(a.out, a.frames, a.size, a.cadrays_root or None)))
        if a.noise_seed:
            make(a.out, a.frames, a.size, a.cadrays_root or None, settings={"default": {}}, seeds=(2,), write_meta=False)
    elif a.cmd == "compare":
        rep = compare(a.kit, a.occt); print_report(rep)
        json.dump(rep, open(os.path.join(a.occt, "pin_report.json"), "w"), indent=1)
    else:
        rep = selfcheck(a.frames, a.size, a.out or None)
        sys.exit(0 if rep["selfcheck"]["default_reproduces_itself"] and rep["selfcheck"]["every_switch_observable"] else 1)


if __name__ == "__main__":
    main()
