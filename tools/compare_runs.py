#!/usr/bin/env python3
"""The comparer half of the reference's regression harness (testing/CADRays_Testing.py), for runs of this backend.

  python tools/compare_runs.py --template DIR --run DIR [--max-diff 2] [--hdr-tol 1e-4] [-u]

A RUN folder holds what `python -m cadrays_amd.run_script script.tcl N --outdir RUN` (or cadrays_headless) writes, the reference's
own layout (main.cxx:199-221): Output_<name>_<n>.txt (first line = frames per second), Output_<name>_<n>.png (BufferDump RGB)
and, from this backend, optionally Output_<name>_<n>.pfm (the linear HDR accumulator).  A TEMPLATE folder is what the
reference's `-u` leaves behind (CADRays_Testing.py:144-167): Result.html with one "File <name>.tcl" / "Framerate = <fps> fps" pair
per script, <name>.png, and here also <name>.pfm.

Per script (CADRays_Testing.py:42-51, 83-84, 226-230):
  * frame rate: (run / template - 1) * 100 within +-max-diff % -> "same"; above -> "faster" (the reference paints it green);
    below -> "slower" (red) -- a regression;
  * LDR image: any non-zero per-pixel difference is written as a white pixel of Diff_<name>.png; a non-empty diff is a failure;
  * HDR image (both sides have a .pfm): relative L2 distance <= hdr-tol (1e-4).
Writes RUN/Result.html (readable by the reference's own parser) and RUN/compare.json, prints the JSON summary, exit status 1 when
any script regressed.  -u promotes RUN to TEMPLATE instead (Result.html + <name>.png / .pfm), like the reference's -u.
"""
import argparse
import datetime
import html.parser
import json
import os
import re
import shutil
import sys

import numpy as np


class ResultParser(html.parser.HTMLParser):
    """the reference's MyHTMLParser (CADRays_Testing.py:20-28): "File x" then "Framerate = y fps" """

    def __init__(self):
        super().__init__()
        self.file, self.result = "", {}

    def handle_data(self, data):
        data = data.strip()
        if data.startswith("File"):
            self.file = data[data.find(" ") + 1:]
        elif data.startswith("Framerate"):
            self.result[self.file] = float(data[data.find("=") + 2: data.find("fps") - 1])


def read_template_rates(template):
    path = os.path.join(template, "Result.html")
    if not os.path.isfile(path):
        return {}
    p = ResultParser()
    p.feed(open(path).read())
    return p.result


def read_png(path):
    from PIL import Image
    return np.asarray(Image.open(path).convert("RGB"))


def write_png(path, arr):
    from PIL import Image
    Image.fromarray(arr).save(path)


def read_pfm(path):
    with open(path, "rb") as f:
        kind = f.readline().strip()
        w, h = (int(x) for x in f.readline().split())
        scale = float(f.readline())
        data = np.frombuffer(f.read(), "<f4" if scale < 0 else ">f4").reshape(h, w, 3 if kind == b"PF" else 1)
    return data[::-1].astype(np.float32)                    # PFM stores the bottom row first


def write_pfm(path, rgb):
    rgb = np.ascontiguousarray(rgb, np.float32)
    with open(path, "wb") as f:
        f.write(b"PF\n%d %d\n-1.0\n" % (rgb.shape[1], rgb.shape[0]))
        f.write(rgb[::-1].astype("<f4").tobytes())


def run_outputs(run):
    """{name: {"n": frames, "fps": float, "png": path or None, "pfm": path or None}} from Output_<name>_<n>.txt"""
    out = {}
    for f in sorted(os.listdir(run)):
        m = re.match(r"Output_(.*)_(\d+)\.txt$", f)
        if not m:
            continue
        base = os.path.join(run, f[:-4])
        try:
            fps = float(open(os.path.join(run, f)).readline().split()[0])
        except (ValueError, IndexError):
            fps = float("nan")
        out[m.group(1)] = {"n": int(m.group(2)), "fps": fps,
                           "png": base + ".png" if os.path.isfile(base + ".png") else None,
                           "pfm": base + ".pfm" if os.path.isfile(base + ".pfm") else None}
    return out


def write_result_html(path, date, rows):
    """rows: (script file name, fps, template fps or None, max_diff, image triple or None) -- the reference's createHTML layout"""
    with open(path, "w") as f:
        f.write("<html><head><title>Result</title></head><body>\n<h1>%s</h1>\n\n<ol>\n" % date.strftime("%d/%m/%Y %H:%M:%S"))
        for name, fps, prev, max_diff, _ in rows:
            f.write('<li>\n<p style="font-size:25px"><strong>File %s</strong></p>\n<ul>\n<li>\n' % name)
            if prev:
                diff = (fps / prev - 1) * 100
                text = "Framerate = %g fps (prev = %g) [%+.4f%%]" % (fps, prev, diff)
                if abs(diff) > max_diff:
                    text = '<span style="background-color:%s">%s</span>' % ("green" if diff > 0 else "red", text)
            else:
                text = "Framerate = %g fps" % fps
            f.write('<p style="font-size:20px"><strong>%s</strong></p>\n</li>\n</ul>\n</li>\n' % text)
        for name, _, _, _, img in rows:
            if not img:
                continue
            f.write('<li>\n<p style="font-size:25px"><strong>File %s</strong></p>\n' % name)
            cols = [("Output result", img[0])] + ([("Model result", img[1]), ("Difference", img[2])] if img[1] else [])
            f.write("<table><tr>%s</tr>\n<tr>%s</tr></table>\n</li>\n" % (
                "".join("<th>%s</th>" % c for c, _ in cols),
                "".join('<td><img src="file://%s" width="100%%" height="100%%"></td>' % p for _, p in cols)))
        f.write("</ol>\n</body></html>\n")


def compare(template, run, max_diff=2.0, hdr_tol=1e-4):
    rates = read_template_rates(template)
    results, rows = [], []
    for name, o in run_outputs(run).items():
        script = name + ".tcl"
        r = {"script": script, "frames": o["n"], "fps": o["fps"], "template_fps": rates.get(script), "fps_diff_pct": None, "fps_status": "no template",
             "ldr_status": "no image", "ldr_diff_pixels": None, "hdr_status": "no image", "hdr_rel_l2": None}
        if r["template_fps"]:
            d = (o["fps"] / r["template_fps"] - 1) * 100
            r["fps_diff_pct"] = round(d, 4)
            r["fps_status"] = "same" if abs(d) <= max_diff else ("faster" if d > 0 else "slower")
        img = None
        if o["png"]:
            tpng = os.path.join(template, name + ".png")
            img = (os.path.abspath(o["png"]), "", "")
            r["ldr_status"] = "no template"
            if os.path.isfile(tpng):
                a, b = read_png(o["png"]), read_png(tpng)
                diff_path = os.path.join(run, "Diff_" + name + ".png")
                if a.shape != b.shape:
                    r["ldr_status"], r["ldr_diff_pixels"] = "size differs", int(a.shape[0] * a.shape[1])
                else:
                    mask = (a != b).any(2)                                   # ImageChops.difference(...).point(0 if x == 0 else 255)
                    r["ldr_diff_pixels"] = int(mask.sum())
                    r["ldr_status"] = "identical" if not mask.any() else "differs"
                    write_png(diff_path, (mask * 255).astype(np.uint8))
                    img = (img[0], os.path.abspath(tpng), os.path.abspath(diff_path))
        if o["pfm"]:
            tpfm = os.path.join(template, name + ".pfm")
            r["hdr_status"] = "no template"
            if os.path.isfile(tpfm):
                a, b = read_pfm(o["pfm"]).astype(np.float64), read_pfm(tpfm).astype(np.float64)
                if a.shape != b.shape:
                    r["hdr_status"] = "size differs"
                else:
                    r["hdr_rel_l2"] = float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))
                    r["hdr_status"] = "within tolerance" if r["hdr_rel_l2"] <= hdr_tol else "differs"
        r["pass"] = r["fps_status"] != "slower" and r["ldr_status"] not in ("differs", "size differs") and r["hdr_status"] not in ("differs", "size differs")
        results.append(r)
        rows.append((script, o["fps"], r["template_fps"], max_diff, img))
    date = datetime.datetime.now()
    write_result_html(os.path.join(run, "Result.html"), date, rows)
    summary = {"template": os.path.abspath(template), "run": os.path.abspath(run), "max_diff_pct": max_diff, "hdr_tol": hdr_tol,
               "scripts": results, "pass": all(r["pass"] for r in results) and bool(results)}
    json.dump(summary, open(os.path.join(run, "compare.json"), "w"), indent=1)
    return summary


def promote(template, run):
    """-u: the run becomes the template (CADRays_Testing.py:144-167)"""
    os.makedirs(template, exist_ok=True)
    outs = run_outputs(run)
    rows = [(name + ".tcl", o["fps"], None, 0.0, None) for name, o in outs.items()]
    write_result_html(os.path.join(template, "Result.html"), datetime.datetime.now(), rows)
    for name, o in outs.items():
        for ext in ("png", "pfm"):
            if o[ext]:
                shutil.copyfile(o[ext], os.path.join(template, name + "." + ext))
    return {"promoted": sorted(outs), "template": os.path.abspath(template)}


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--template", "-m", required=True)
    ap.add_argument("--run", "-o", required=True)
    ap.add_argument("--max-diff", "-d", type=float, default=2.0, help="frame-rate tolerance in percent (reference default 2)")
    ap.add_argument("--hdr-tol", type=float, default=1e-4)
    ap.add_argument("-u", "--update", action="store_true", help="promote the run to the template")
    a = ap.parse_args(argv)
    if not os.path.isdir(a.run):
        sys.exit("run folder not found: " + a.run)
    if a.update:
        print(json.dumps(promote(a.template, a.run)))
        return 0
    if not os.path.isdir(a.template):
        sys.exit("template folder not found: " + a.template)
    s = compare(a.template, a.run, a.max_diff, a.hdr_tol)
    print(json.dumps(s))
    return 0 if s["pass"] else 1


if __name__ == "__main__":
    sys.exit(main())
