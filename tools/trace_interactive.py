"""24 free-running Redraw()s at look-ahead 1 on C3 (the reference regime, AppViewer.cxx:1045-1047): the workload profiles/r3/interactive_counters.txt was
collected on -- rocprofv3 --kernel-trace / --pmc ... --output-format csv -- python3 tools/trace_interactive.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
from cadrays_amd import scenes
from cadrays_amd.view import View
v = View(0).load_scene(scenes.baseline_config("C3"))
v.set_lookahead(1); v.reset()
for _ in range(8): v.Redraw()
v.sync()
t0 = time.perf_counter()
for _ in range(24): v.Redraw()
v.sync()
print("redraw/s", 24 / (time.perf_counter() - t0))
