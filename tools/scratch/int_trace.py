import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
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
