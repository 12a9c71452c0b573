#!/usr/bin/env python3
"""bench.py -- headline benchmark of the path-tracing hot path.

  python bench.py --gpus N --steps K --warmup W

A "step" = one crh_render pass of `--spp` (default: one full 256 M-path batch = 128 samples per pixel at 1080p) over the rank's tiles of the workload
(BASELINE.json config C3: 1 M random triangles, glass + glossy double-layer BSDFs, HDR sky, 1080p).
At N > 1 (one process per GPU under torch.distributed.run) tiles are interleaved across ranks, every rank
renders spp*N samples of its tiles per step (fixed per-GPU work -> weak scaling), and each step ends with
the RCCL reduce of the float4 framebuffer to rank 0.  Inputs are resident in HBM before the timed region.

Prints ONE JSON line (rank 0): metric Mrays/s (nearest-hit + any-hit rays actually traced, whole job),
plus `roofline` for the dominant kernel (k_trace_nearest; HIP-event kernel time measured inside the timed
region, algorithmic bytes from the deterministic counters) and `cpu_baseline` (the CPU oracle timed on this
box's cores on a bounded tile sample of the same workload).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8 TB/s peak


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="C3", choices=["C1", "C2", "C3", "C5"])
    ap.add_argument("--spp", type=int, default=0, help="samples per pixel per step (per GPU share); 0 = what fills one 256 M-path batch (128 at 1080p, 32 at 4K)")
    ap.add_argument("--tris", type=int, default=0, help="override triangle count (debug)")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--pmc-traffic", type=float, default=None, help="HBM bytes per launch from a separate rocprofv3 --pmc pass")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    import torch
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("CRH_BENCH_BACKEND", "nccl")     # "gloo" + CRH_BENCH_SHARE_DEVICE=1: rehearsal of the N > 1 flow on one GPU
        if os.environ.get("CRH_BENCH_SHARE_DEVICE") == "1":
            local = 0
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
        else:
            dist.init_process_group(backend)
    n_gpus = world

    from cadrays_amd import scenes, sharding
    from cadrays_amd.view import View

    sc = scenes.baseline_config(args.config, args.width or None, args.height or None, args.tris or None)
    t0 = time.time()
    v = View(local).load_scene(sc)
    build_s = time.time() - t0
    fb = sharding.DeviceFramebuffer(v) if world > 1 else None
    tiles = sharding.tiles_for_rank(v.n_tiles(), rank, world)
    if args.spp <= 0:                     # one full path batch per step: 2^28 slots / (tiles x 32 x 32 pixels)
        args.spp = max(1, (256 << 20) // (v.n_tiles() * sc.params.tile_size ** 2))
    spp_step = args.spp * world          # fixed per-GPU work: 1/N of the tiles, N x the samples

    def barrier():
        v.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def step(i):
        v.render_tiles(tiles, i * spp_step, spp_step)
        if dist is not None:
            v.sync()                                          # the accumulator is written on the context's own stream
            sharding.reduce_framebuffer(fb.tensor, 0)         # RCCL reduce of a staging copy; returns synchronised

    if dist is not None:                                      # RCCL builds its rings / channels on first use: keep that out of the
        sharding.reduce_framebuffer(fb.tensor, 0)             # timed steps even when the caller asks for --warmup 0
    for i in range(args.warmup):
        step(i)
    barrier()
    v.reset()
    v.enable_kernel_timing(True)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    barrier()
    dt = time.perf_counter() - t0
    v.enable_kernel_timing(False)
    kt = v.kernel_timing()
    st = v.stats()
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=f"cuda:{local}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        cnt = torch.tensor([st["rays_nearest"], st["rays_any"], st["samples"]], dtype=torch.float64, device=f"cuda:{local}")
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        rays_n, rays_a, samples = (float(x) for x in cnt.tolist())
    else:
        rays_n, rays_a, samples = float(st["rays_nearest"]), float(st["rays_any"]), float(st["samples"])

    # ---- counting pass (untimed): the same frames again with node / triangle counters on (rank 0's shard)
    roof = None
    if rank == 0:
        v.reset(); v.enable_counters(True)
        for i in range(args.steps):
            v.render_tiles(tiles, (args.warmup + i) * spp_step, spp_step)
        cs = v.stats()
        v.enable_counters(False)
        assert cs["rays_nearest"] == st["rays_nearest"], "counting pass traced different rays"
        launches = max(kt["trace_nearest_launches"], 1)
        # algorithmic bytes of k_trace_nearest: 3 x float4 of node per inner visit, 3 x float4 per triangle test, 32-B ray + 16-B hit per ray
        alg_bytes = 48.0 * cs["nodes_nearest"] + 48.0 * cs["tris_nearest"] + 48.0 * cs["rays_nearest"]
        per_launch = alg_bytes / launches
        avg_ms = kt["trace_nearest_ms_total"] / launches
        achieved = per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        traffic = args.pmc_traffic
        traffic_source = "--pmc-traffic" if traffic is not None else None
        pmc_file = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if traffic is None and args.config == "C3" and not (args.tris or args.width or args.height) and os.path.exists(pmc_file):
            # HBM-side bytes per launch of this kernel from the committed rocprofv3 --pmc passes of this same command
            pmc = json.load(open(pmc_file))
            traffic = pmc.get("hbm_bytes_per_launch")
            traffic_source = "profiles/pmc_traffic.json (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command)"
            if traffic is not None and args.spp != pmc.get("spp_per_step", 32):
                traffic = traffic * args.spp / pmc.get("spp_per_step", 32)
        roof = {"bound": "hbm", "kernel": "k_trace_nearest", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic, "traffic_source": traffic_source,
                "alg_bytes_per_launch": round(per_launch), "avg_launch_ms": round(avg_ms, 4), "launches": int(launches),
                "kernel_time_share": round(kt["trace_nearest_ms_total"] / max(kt["render_ms_total"], 1e-9), 3),
                "nodes_per_ray": round(cs["nodes_nearest"] / max(cs["rays_nearest"], 1), 2),
                "tris_per_ray": round(cs["tris_nearest"] / max(cs["rays_nearest"], 1), 2)}

    # ---- CPU baseline: the oracle (a port, not the reference: OCCT has no CPU path tracer) on this box's cores
    cpu = None
    if rank == 0 and not args.no_cpu and world == 1:          # reported on rank 0 at N = 1 only
        from oracle.pyoracle import Oracle
        ncores = os.cpu_count() or 1
        Oracle.set_threads(ncores)
        o = Oracle().load_scene(sc)
        nt = o.n_tiles()
        sample = np.unique(np.linspace(0, nt - 1, 32).astype(np.uint32))
        o.render_tiles(sample, 0, 1)                      # calibration (also warms the threads)
        s0 = o.stats()
        rate = (s0["rays_nearest"] + s0["rays_any"]) / max(s0["seconds"], 1e-9)
        per_tile = (s0["rays_nearest"] + s0["rays_any"]) / len(sample)
        want = int(min(nt, max(32, rate * args.cpu_seconds / max(per_tile, 1))))
        sample = np.unique(np.linspace(0, nt - 1, want).astype(np.uint32))
        # first pass at 1 spp re-measures the rate with all threads busy; then size spp for ~cpu_seconds
        o.reset(); o.render_tiles(sample, 0, 1)
        s1 = o.stats()
        rate = (s1["rays_nearest"] + s1["rays_any"]) / max(s1["seconds"], 1e-9)
        spp_cpu = int(max(1, min(64, rate * args.cpu_seconds / max(s1["rays_nearest"] + s1["rays_any"], 1))))
        o.reset(); o.render_tiles(sample, 0, spp_cpu)
        s1 = o.stats()
        cpu = {"value": round((s1["rays_nearest"] + s1["rays_any"]) / s1["seconds"] / 1e6, 3), "unit": "Mrays/s", "cores": ncores,
               "kind": "port", "sample": f"{len(sample)} of {nt} 32x32 tiles of the same workload, {spp_cpu} spp, {s1['seconds']:.1f} s, OpenMP oracle"}

    if rank == 0:
        mrays = (rays_n + rays_a) / dt / 1e6
        out = {
            "metric": "Mrays/s", "value": round(mrays, 2), "unit": "Mrays/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.config}: {len(sc.tri)} random triangles, {len(sc.materials)} BSDF(s), "
                                   f"{'HDR sky env' if sc.env is not None else 'constant env'}, {len(sc.lights)} light(s), "
                                   f"{sc.params.width}x{sc.params.height}, depth {sc.params.max_depth}",
                       "spp_per_step": spp_step, "tiles_per_rank": int(len(tiles)), "parallelism": f"tiles x{n_gpus} + RCCL reduce" if n_gpus > 1 else "single GPU",
                       "msamples_per_s": round(samples / dt / 1e6, 3), "rays_nearest": int(rays_n), "rays_any": int(rays_a),
                       "build_upload_s": round(build_s, 2), "host_cores": os.cpu_count()},
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()                                       # rank 0 may still be in its counting pass
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
