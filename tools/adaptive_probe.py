import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
from cadrays_amd import scenes
from cadrays_amd.view import View
v = View(0).load_scene(scenes.baseline_config("C3"))
v.set_adaptive(True, 512)
for _ in range(40):
    v.Redraw()
v.sync()
