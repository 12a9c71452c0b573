#!/usr/bin/env python3
"""The application's own call pattern, measured (verdict r4 item 1): CADRays calls `View->Redraw()` once per GUI frame = +1 sample per pixel
(AppViewer.cxx:1045-1047), shows every frame (ImGui::Image, AppViewer.cxx:1099) and restarts the accumulation on every camera change
(AppViewer.cxx:979-984).  Four figures on a BASELINE config at its full resolution:

  first_frame_after_a_restart_ms   crh_reset + crh_render(1) + crh_sync, median / min of --trials (what the user waits for after letting go of the mouse)
  drag_frames_per_s                every frame: crh_set_camera (a slightly turned eye) + crh_reset + crh_render(1) + asynchronous LDR read-back, collected
                                   two frames later (orbiting the model with the mouse held down)
  displayed_frames_per_s           every frame: crh_render(1) + asynchronous LDR read-back two frames behind (a still camera, every frame on screen)
  free_running_redraw_per_s        crh_render(1) back to back, nothing read (the figure round 3 / 4 quoted)

  python tools/bench_redraw.py [--config C3] [--frames 128] [--trials 15]         (CRH_LANES / CRH_FRAME_KERNEL / CRH_PIPE_DEPTH select schedules)"""
import argparse, dataclasses, json, math, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")          # the host exports it before the first HIP call (crh_query_pipeline_capacity)


def drag_camera(cam0, i):
    """frame i of the drag: the eye orbits the scene centre, looking at it"""
    a = 0.002 * i
    r = math.sqrt(sum(x * x for x in cam0.eye))
    eye = (r * math.sin(a), -r * math.cos(a), 0.0)
    d = tuple(-x / r for x in eye)
    return dataclasses.replace(cam0, eye=eye, dir=d)


def measure(v, cam0, frames=128, trials=15):
    out = {}
    # the first 30 frame-kernel frames after a build are the library's measurement of its feeder count (crh_get_frame_tuning): let it settle first
    def settled():
        ft = v.frame_tuning(); v.tile_order()
        return (not ft["enabled"] or ft["feeders"]) and v.tile_order_calls["verdict"] != 0
    for _ in range(96):                                         # (the feeder count, then whether the sorted tile list pays on this scene: crh_get_tile_order)
        if settled(): break
        v.reset(); v.Redraw(); v.sync()
    out["frame_tuning"] = v.frame_tuning(); out["tile_order"] = dict(v.tile_order_calls)
    # ---- a lone frame after a restart
    ts = []
    for _ in range(trials):
        v.reset(); v.sync()
        t = time.perf_counter()
        v.Redraw(); v.sync()
        ts.append((time.perf_counter() - t) * 1e3)
    out["first_frame_after_a_restart_ms"] = round(statistics.median(ts), 3)
    out["first_frame_after_a_restart_ms_min"] = round(min(ts), 3)
    out["tile_list_replaced_up_to_the_end_of_the_lone_frames"] = v.tile_order()[1]; out["tile_order_calls_up_to_the_end_of_the_lone_frames"] = dict(v.tile_order_calls)

    import numpy as np
    shown = np.empty((v.height, v.width, 3), np.uint8)          # the host's one staging buffer (INTEGRATION.md `myLdr`, cadrays_headless.cpp `shown`)
    # ---- the drag: camera change -> restart -> one frame -> shown
    def cam(i):
        return drag_camera(cam0, i)
    for loop in ("warm", "timed"):
        n = 8 if loop == "warm" else frames
        v.sync()
        t = time.perf_counter()
        for i in range(n):
            v.set_camera(cam(i)); v.reset(); v.Redraw()
            if i >= 2: v.read_ldr_end(shown)
            v.read_ldr_begin()
        v.read_ldr_end(shown); v.read_ldr_end(shown); v.sync()
        dt = time.perf_counter() - t
    out["drag_frames_per_s"] = round(frames / dt, 1)
    out["tile_list_replaced_up_to_the_end_of_the_drag"] = v.tile_order()[1]; out["tile_order_calls_up_to_the_end_of_the_drag"] = dict(v.tile_order_calls)
    v.set_camera(cam0); v.reset()

    # ---- a still camera, every frame displayed
    for loop in ("warm", "timed"):
        n = 8 if loop == "warm" else frames
        v.sync()
        t = time.perf_counter()
        for i in range(n):
            v.Redraw()
            if i >= 2: v.read_ldr_end(shown)
            v.read_ldr_begin()
        v.read_ldr_end(shown); v.read_ldr_end(shown); v.sync()
        dt = time.perf_counter() - t
    out["displayed_frames_per_s"] = round(frames / dt, 1)

    # ---- free-running
    v.reset()
    for _ in range(16): v.Redraw()
    v.sync()
    t = time.perf_counter()
    for _ in range(frames): v.Redraw()
    v.sync()
    out["free_running_redraw_per_s"] = round(frames / (time.perf_counter() - t), 1)
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C3")
    ap.add_argument("--frames", type=int, default=128)
    ap.add_argument("--trials", type=int, default=15)
    ap.add_argument("--no-torch", action="store_true", help="do not import torch first: the process then runs on the HIP runtime the library was linked against (/opt/rocm), "
                    "not on the one bundled with the torch wheel (round 6: the two differ in what a lone frame costs the host)")
    a = ap.parse_args()
    if not a.no_torch:
        import torch  # noqa: F401
    from cadrays_amd import scenes
    from cadrays_amd.view import View
    sc = scenes.baseline_config(a.config)
    v = View(0).load_scene(sc)
    out = {"config": a.config, "env": {k: os.environ[k] for k in sorted(os.environ) if k.startswith(("CRH_", "ROC_", "HSA_")) or k == "GPU_MAX_HW_QUEUES"},
           "hip_runtime": [l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l][:1]}
    out.update(measure(v, sc.camera, a.frames, a.trials))
    st = v.stats()
    out["rays_per_frame"] = None
    v.reset(); v.Redraw(); v.sync()
    st = v.stats()
    out["rays_per_frame"] = int(st["rays_nearest"] + st["rays_any"])
    print(json.dumps(out), flush=True)
