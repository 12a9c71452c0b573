import os, sys, time, json
sys.path.insert(0, '.'); sys.path.insert(0, 'tools')
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np, torch
from cadrays_amd import scenes
from cadrays_amd.view import View
import bench_redraw
sc = scenes.baseline_config("C3"); v = View(0).load_scene(sc)
what = os.environ.get("PROBE_BEFORE", "")
if what == "torch_sync": torch.cuda.synchronize()
elif what == "torch_tensor": torch.zeros(4, device="cuda:0"); torch.cuda.synchronize()
elif what == "torch_event": e = torch.cuda.Event(enable_timing=True); e.record(); torch.cuda.synchronize()
shown = np.empty((v.height, v.width, 3), np.uint8)
res = []
for rep in range(3):
    for loop in ("warm", "timed"):
        n = 8 if loop == "warm" else 128
        v.sync(); t = time.perf_counter()
        for i in range(n):
            v.set_camera(bench_redraw.drag_camera(sc.camera, i)); v.reset(); v.Redraw()
            if i >= 2: v.read_ldr_end(shown)
            v.read_ldr_begin()
        v.read_ldr_end(shown); v.read_ldr_end(shown); v.sync(); dt = time.perf_counter() - t
    res.append(round(128 / dt, 1))
o, n = v.tile_order()
print(json.dumps({"before": what, "env": os.environ.get("CRH_TILE_ORDER"), "drag_only": res, "replaced": n, "calls": v.tile_order_calls}))
