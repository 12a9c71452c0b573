#!/usr/bin/env python3
"""Soak: tens of thousands of API calls in a GUI-like loop (Redraw, asynchronous read-back, camera / material / transform edits, adaptive
switches, resets) on one context; device memory, host memory and the frame rate must not drift (event pools, staging buffers,
read-back slots are reused, not grown).   python tools/soak.py [frames]

What "must not drift" means for host memory (round 6): the FIRST launch of a kernel family -- the two-level kernels when the first object moves, the
patched-record kernels when the first object is erased, ... -- costs the HIP runtime about 189 MiB of anonymous host memory, once (four such steps in this
call mix, then none in 250 000 further calls: profiles/r6/soak_host_memory_steps.txt).  So the criterion is: nothing grows over the SECOND HALF of the run;
the one-time steps are reported (`host_one_time_steps_MiB`)."""
import dataclasses, json, os, resource, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cadrays_amd import scenes
from cadrays_amd.view import View

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 60000
sc = scenes.cornell_box(True, 320, 240)
tri_obj = sc.tri[:, 3].astype(np.int32); nO = int(tri_obj.max()) + 1
xf = np.tile(np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], np.float32), (nO, 1))
sc = dataclasses.replace(sc, tri_object=tri_obj, obj_xform=xf)
v = View(0).load_scene(sc)
r = np.random.default_rng(1)


def mem():
    free, total = torch.cuda.mem_get_info(0)
    return (total - free) / 2**20, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024


def rss_now():
    return int(open("/proc/self/statm").read().split()[1]) * os.sysconf("SC_PAGE_SIZE") / 2**20


def burst(n):
    inflight = 0
    t0 = time.perf_counter()
    for i in range(n):
        k = r.integers(0, 40)
        if k == 0: v.set_camera(dataclasses.replace(sc.camera, eye=(0.5 + 0.1 * r.normal(), -1.5, 0.5))); v.reset()
        elif k == 1:
            m = xf.copy(); m[3, 3::4] += (0.02 * r.normal(size=3)).astype(np.float32); v.set_transforms(m)
        elif k == 2:
            mats = list(sc.materials); mats[0] = dataclasses.replace(mats[0], Kd=np.float32(r.random(3))); v.set_materials(mats)
        elif k == 3: v.set_adaptive(bool(r.integers(0, 2)), 32)
        elif k == 4: v.set_lookahead(int(r.choice([1, 4])))
        elif k == 5 and not os.environ.get("SOAK_NO_VISIBILITY"): v.set_visibility((r.random(nO) > 0.2).astype(np.uint8))            # the eye icons of the scene tree (round 6)
        v.Redraw()
        if inflight == 2: v.read_ldr_end(); inflight -= 1
        v.read_ldr_begin(); inflight += 1
    while inflight: v.read_ldr_end(); inflight -= 1
    v.sync()
    return n / (time.perf_counter() - t0)


burst(2000)
d0, h0 = mem(); rates = []; rss = [round(rss_now(), 1)]
for _ in range(max(1, frames // 5000)):
    rates.append(round(burst(5000), 1)); rss.append(round(rss_now(), 1))
d1, h1 = mem()
out = {"frames": 5000 * len(rates), "redraw_per_s_per_5000": rates, "device_MiB_before": round(d0, 1), "device_MiB_after": round(d1, 1),
       "host_rss_MiB_per_5000": rss, "host_maxrss_MiB_before": round(h0, 1), "host_maxrss_MiB_after": round(h1, 1), "finite": bool(np.isfinite(v.read_hdr()).all())}
half = len(rss) // 2
out["host_one_time_steps_MiB"] = [round(b - a, 1) for a, b in zip(rss, rss[1:]) if b - a > 64]
out["host_growth_second_half_MiB"] = round(rss[-1] - rss[half], 1)
out["pass"] = abs(d1 - d0) < 64 and rss[-1] - rss[half] < 64 and min(rates) > 0.7 * max(rates) and out["finite"]
print(json.dumps(out))
sys.exit(0 if out["pass"] else 1)
