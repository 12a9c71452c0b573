#!/usr/bin/env python3
"""The reference's regression harness, whole (testing/CADRays_Testing.py, Python 2 + Windows `start /wait cmd /c`), for this backend:
same options, same folder protocol, same Result.html, Python 3, any OS.

  python tools/cadrays_testing.py -i SCRIPTS -m MODEL [-o OUTPUT] [-c CADRAYS] [-f FRAMES] [-d MAXDIFF]      run + compare
  python tools/cadrays_testing.py -u -o OUTPUT -m MODEL                                                     promote the latest run

  -i  folder with the .tcl scripts; every script is run as  <CADRAYS> <script> <FRAMES>  (CADRays_Testing.py:177-185)
  -c  the program to run; default: this backend's script host,  python -m cadrays_amd.run_script  (the reference's CADRays.exe
      takes the same two arguments, main.cxx:164-189, and writes Output_<name>_<n>.png / .txt next to the script)
  -f  frames per script (default 100)            -d  frame-rate tolerance in percent (default 2)
  -o  output folder (default: the scripts folder): the run lands in  OUTPUT/<dd_mm_YYYY HH_MM_SS>/  with Result.html and Diff_*.png
  -m  model (template) folder: Result.html + <name>.png of the accepted run;  -u copies the LATEST dated run of OUTPUT into it

What the reference does and this does too (line numbers of CADRays_Testing.py): stale Output_* files are removed from the scripts
folder (170-173); after the runs the images move into the dated folder (205-207), frame rates are read from the first line of each
Output_*.txt (213-216), images are compared pixel by pixel into Diff_<name>.png (226-230), Result.html lists "File x" / "Framerate =
y fps (prev = z) [d%]" with red / green marks beyond the tolerance (42-51), the .txt files are deleted (238-241).  Beyond the
reference: the linear HDR image (.pfm, when the host wrote one) travels and is compared too, a JSON summary is printed, and the exit
status is 1 when a script got slower than the tolerance or an image differs (the reference only paints the page)."""
import datetime
import getopt
import json
import os
import re
import shutil
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import compare_runs  # noqa: E402

DATE_FMT = "%d_%m_%Y %H_%M_%S"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def latest_run(output):
    best = None
    for d in os.listdir(output):
        if os.path.isdir(os.path.join(output, d)) and re.match(r"(\d\d?)_(\d\d?)_(\d\d\d\d) (\d\d?)_(\d\d?)_(\d\d?)$", d):
            t = datetime.datetime.strptime(d, DATE_FMT)
            if best is None or t > best:
                best = t
    return best


def main(argv=None):
    try:
        opts, _ = getopt.getopt(sys.argv[1:] if argv is None else argv, "hi:c:f:d:o:m:u")
    except getopt.GetoptError:
        print(__doc__); return 2
    inp = out = model = ""; cadrays = None; frames = 100; max_diff = 2.0; update = False
    for o, a in opts:
        if o == "-h": print(__doc__); return 0
        elif o == "-u": update = True
        elif o == "-i": inp = a
        elif o == "-o": out = a
        elif o == "-c": cadrays = a
        elif o == "-m": model = a
        elif o == "-f": frames = int(a)
        elif o == "-d": max_diff = float(a)
    if not update and not os.path.isdir(inp): print("Path to scripts folder is incorrect"); return 2
    if not update and cadrays is not None and not os.path.isfile(cadrays): print("Path to the CADRays program is incorrect"); return 2
    if not model or not os.path.isdir(model): print("Path to the folder with results for comparing is incorrect"); return 2
    if not out:
        if update: print("Path to output folder is incorrect"); return 2
        out = inp
    elif not os.path.isdir(out): print("Path to output folder is incorrect"); return 2
    out, model = os.path.abspath(out), os.path.abspath(model)

    if update:
        t = latest_run(out)
        if t is None: print("No results founded"); return 2
        run = os.path.join(out, t.strftime(DATE_FMT))
        rates = compare_runs.read_template_rates(run)             # the run's own Result.html holds the rates (its .txt files are gone), 152-157
        compare_runs.write_result_html(os.path.join(model, "Result.html"), t, [(name, fps, None, 0.0, None) for name, fps in rates.items()])
        copied = []
        for f in sorted(os.listdir(run)):
            m = re.match(r"Output_(.*)_(\d+)\.(png|pfm)$", f)
            if m: shutil.copyfile(os.path.join(run, f), os.path.join(model, m.group(1) + "." + m.group(3))); copied.append(f)
        print(json.dumps({"promoted": t.strftime(DATE_FMT), "rates": rates, "images": copied, "model": model}))
        return 0

    inp = os.path.abspath(inp)
    for f in os.listdir(inp):
        if re.match(r"Output_(.*)\.((txt)|(png)|(pfm))$", f): os.remove(os.path.join(inp, f))
    scripts = sorted(f for f in os.listdir(inp) if os.path.isfile(os.path.join(inp, f)) and os.path.splitext(f)[1].upper() == ".TCL")
    failed = []
    for s in scripts:
        if cadrays is None:
            cmd = [sys.executable, "-m", "cadrays_amd.run_script", os.path.join(inp, s), str(frames), "--outdir", inp, "--hdr"]
            p = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True)
        else:
            p = subprocess.run([os.path.abspath(cadrays), os.path.join(inp, s), str(frames)], cwd=inp, capture_output=True, text=True)
        if p.returncode or not os.path.isfile(os.path.join(inp, "Output_%s_%d.txt" % (os.path.splitext(s)[0], frames))):
            failed.append({"script": s, "returncode": p.returncode, "stderr": p.stderr.strip()[-400:]})
    date = datetime.datetime.now()
    run = os.path.join(out, date.strftime(DATE_FMT))
    os.makedirs(run, exist_ok=True)
    for f in os.listdir(inp):
        if re.match(r"Output_(.*)\.((txt)|(png)|(pfm))$", f): shutil.move(os.path.join(inp, f), os.path.join(run, f))
    summary = compare_runs.compare(model, run, max_diff)
    summary["did_not_run"] = failed
    summary["pass"] = bool(summary["scripts"]) and not failed and all(r["pass"] for r in summary["scripts"])
    json.dump(summary, open(os.path.join(run, "compare.json"), "w"), indent=1)
    for f in os.listdir(run):                                       # the reference deletes the rate files once Result.html holds them
        if re.match(r"Output_(.*)\.txt$", f): os.remove(os.path.join(run, f))
    print(json.dumps(summary))
    return 0 if summary["pass"] or not os.path.isfile(os.path.join(model, "Result.html")) else 1


if __name__ == "__main__":
    sys.exit(main())
