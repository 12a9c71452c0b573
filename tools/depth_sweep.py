import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch
from cadrays_amd import scenes, abi
from cadrays_amd.view import View
sc = scenes.baseline_config("C3")
v = View(0).load_scene(sc)
for depth in (1, 2, 3, 4, 6, 8, 10):
    row = []
    for mode in (abi.SCHEDULE_AUTO, abi.SCHEDULE_STAGED):
        v.ChangeRenderingParams(max_depth=depth); v.set_schedule(mode)
        ts = []
        for _ in range(7):
            v.reset(); v.sync(); t = time.perf_counter(); v.Redraw(); v.sync(); ts.append((time.perf_counter() - t) * 1e3)
        st = v.stats()
        row.append(statistics.median(ts))
    print(f"depth {depth:2d}: frame kernel {row[0]:.3f} ms   staged {row[1]:.3f} ms   rays {st['rays_nearest']}", flush=True)
