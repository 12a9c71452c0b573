#!/usr/bin/env python3
"""bench.py -- headline benchmark of the path-tracing hot path.

  python bench.py --gpus N --steps K --warmup W [--config C3|C2|C4|C5|C1] [--scaling weak|strong]

A "step" = one crh_render_tiles pass over the rank's tiles of the workload (default BASELINE.json config C3: 1 M random
triangles, glass + glossy double-layer BSDFs, HDR sky, 1080p; `--spp` samples per pixel per step, default 512 = two batches
of 1024 tiles x 512 samples: the library cuts a call into batches of <= 2^29 paths).  Inputs are resident in HBM before the timed region.

N = 1, default workload: after the headline leg the OTHER single-GPU configs of BASELINE.json -- C5 (10 M triangles at 4K: the one whose scene does
not fit the caches, i.e. where HBM is the roof), C2, C1 -- run as short legs (1 warm-up + 4 timed steps) in the same process, each with its own parity
gates and roofline: config.other_configs_timed.  `value` stays the headline config.  (--other-configs none switches them off.)

N > 1: one process per GPU.  Under `python -m torch.distributed.run` (WORLD_SIZE set) this process is one of the ranks;
started by hand as `python bench.py --gpus N` it SPAWNS the N rank processes itself -- before torch or the HIP runtime is
touched in the parent, which only waits for them -- so the advertised command measures N GPUs.  The BVH is built ONCE (rank 0) and handed to the other
ranks (crh_build_prebuilt).  Tiles are interleaved across ranks (k-th tile of the Z-order curve -> rank k mod N, cadrays_amd/sharding.py), no
collective while rendering, and every step ends with the exchange step that assembles the float4 framebuffer on rank 0: the RCCL reduce of whole
frames or the gather of owned tiles (1 / N of the bytes), whichever is faster on the fabric (timed before the run; --assemble forces one).  The line
then carries per-rank render / exchange times, the shard balance and both exchange timings (config.per_rank, .assemble, .scene_hand_over).
  --scaling weak   (default) per-GPU work fixed: every rank renders spp*N samples of its 1/N of the tiles per step
  --scaling strong the job is fixed: every rank renders spp samples of its 1/N of the tiles; `--config C4` = C3's scene,
                   4096 spp per step, strong (BASELINE.json configs[3])

Prints ONE JSON line (rank 0): metric Mrays/s (nearest-hit + any-hit rays actually traced, whole job), `roofline` for the
dominant kernel (k_trace_nearest: HIP-event kernel time measured inside the timed region, algorithmic bytes from the
deterministic counters, memory-side traffic from rocprofv3 --pmc passes THIS RUN starts as child processes after its timed region -- FETCH_SIZE and
WRITE_SIZE in separate passes of a 2-step run of the same workload, --pmc combined with --kernel-trace only (live_traffic; --live-traffic off, a missing
rocprofv3 or a profiler around this process: the committed passes of profiles/pmc_traffic.json, used only when they belong to THIS build)), `cpu_baseline` (the CPU
oracle timed on this box's cores on a bounded tile sample of the same workload) and two parity gates: `parity` (the oracle re-renders sampled tiles of
the WHOLE timed region) and `parity_step0` (>= 32 tiles of the first timed step's samples, replayed untimed with the same schedule).  A differing
pixel ends the run with exit code 1; a checker that could not run is reported as `checker_errors` -- and when NO gate of the headline leg produced a verdict the
run ends with exit code 3 (--allow-unchecked accepts the unchecked number).  Every leg's rate, fractions and gate result, the interactive figures and the measured
ceilings are also flat scalar keys (top level, `config`, `roofline`) and the last key `legs_summary` (flatten_line).
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s peak, ~6.3 TB/s achievable (float4 copy); Infinity Cache 256 MiB
HBM_PEAK_GBPS = 8000.0
HBM_ACHIEVABLE_GBPS = 6300.0
INFINITY_CACHE_BYTES = 256 << 20
NODE_BYTES_FETCHED = 48          # 3 x dwordx4 of the 64-B-stride node are fetched per inner visit (SURVEY 8d assumed 128 B)
PMC_FILE = os.path.join(ROOT, "profiles", "pmc_traffic.json")
PMC_HASH_GLOBS = ("cadrays_amd/csrc/*.hip", "cadrays_amd/csrc/*.cpp", "cadrays_amd/csrc/*.h", "include/*.h")


def pmc_hash_files():
    """Every source that goes into libcadrays_hip.so: the bytes a launch moves depend on the kernels, the node format, the grids and
    batch shapes of crh_api.cpp, the stack sizes of device_types.h and the tree the builder produces."""
    import glob
    out = []
    for g in PMC_HASH_GLOBS:
        out += sorted(os.path.relpath(f, ROOT) for f in glob.glob(os.path.join(ROOT, g)))
    return out


def kernel_source_hash():
    """Identity of the build a PMC measurement belongs to."""
    h = hashlib.sha256()
    for rel in pmc_hash_files():
        h.update(rel.encode())
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def pmc_entry(config, spp, modified):
    """(entry, reason): the committed counter figures of `config` if they were taken on this build of the kernel."""
    if modified:
        return None, "workload overridden on the command line (--tris/--width/--height): no PMC pass of this command"
    if not os.path.exists(PMC_FILE):
        return None, "profiles/pmc_traffic.json missing"
    pmc = json.load(open(PMC_FILE))
    ent = (pmc.get("configs") or {}).get(config)
    if ent is None:
        return None, f"profiles/pmc_traffic.json has no entry for {config}"
    have, want = ent.get("source_hash"), kernel_source_hash()
    if have != want:
        return None, f"stale: PMC passes were taken on kernel build {have}, this is {want} (re-run profiles/pmc_collect.sh)"
    if ent.get("spp_per_step") != spp:
        return None, f"PMC passes used {ent.get('spp_per_step')} spp per step, this run {spp}"
    return ent, None


def being_profiled():
    """rocprofv3 preloads its tool library into the program it runs: a profiler inside a profiled process is not attempted"""
    return any(k.startswith(("ROCPROF", "ROCP_", "ROCPROFILER")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")


def live_traffic(config, timeout_s=240.0):
    """Memory-side counters of the traversal launches taken IN THIS RUN (verdict r3 weak 5): the command the committed passes use (profiles/pmc_collect.sh),
    as a child process under `rocprofv3 --pmc X --kernel-trace` -- one pass per counter, as /opt/skills/guides/MI355X_MICROARCH.md prescribes, --pmc combined with
    nothing but the kernel trace, the program itself right after `--`.  Returns per-launch figures, or {"error": why}."""
    import shutil
    import subprocess
    import tempfile
    global _LIVE_FAILED
    if _LIVE_FAILED:                 # one failed attempt per run: the later legs do not wait for the same time-out again
        return {"error": "not attempted: " + _LIVE_FAILED}
    r = _live_traffic(config, timeout_s, shutil, subprocess, tempfile)
    if "error" in r:
        _LIVE_FAILED = r["error"]
    return r


_LIVE_FAILED = None


def _live_traffic(config, timeout_s, shutil, subprocess, tempfile):
    prof = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if prof is None:
        return {"error": "rocprofv3 not found"}
    if being_profiled():
        return {"error": "this process is itself running under a profiler"}
    sys.path.insert(0, os.path.join(ROOT, "profiles"))
    import pmc_fold
    t0 = time.time()
    child = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", config, "--steps", "2", "--warmup", "1", "--no-cpu", "--no-interactive", "--no-parity",
             "--other-configs", "none", "--live-traffic", "off"]
    vals, launches, tmp = {}, {}, tempfile.mkdtemp(prefix="crh_pmc_")
    env = dict(os.environ, TMPDIR=tmp)
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, ctr)
            try:
                pr = subprocess.run([prof, "--pmc", ctr, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--"] + child,
                                    cwd=tmp, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout_s)
            except subprocess.TimeoutExpired:
                return {"error": f"the {ctr} pass did not finish in {timeout_s:.0f} s"}
            v, n = pmc_fold.counters(d)
            if pr.returncode != 0 or ctr not in v:
                return {"error": f"the {ctr} pass failed (rc {pr.returncode}): " + pr.stdout.decode(errors="replace")[-300:].replace("\n", " | ")}
            vals[ctr], launches[ctr] = v[ctr], n[ctr]
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return {"fetch_size_kb_per_launch": vals["FETCH_SIZE"], "write_size_kb_per_launch": vals["WRITE_SIZE"],
            "hbm_bytes_per_launch": vals["FETCH_SIZE"] * 1024 * 2 + vals["WRITE_SIZE"] * 1024, "launches_per_pass": launches["FETCH_SIZE"],
            "seconds": round(time.time() - t0, 1),
            "how": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (two passes, each with --kernel-trace only) around `bench.py --config %s --steps 2 --warmup 1 --no-cpu "
                   "--no-interactive --no-parity --other-configs none`, started by THIS run as child processes after its timed region; per launch of the timed traversal "
                   "instantiations (k_trace_packets + fall-back pass, k_trace_nearest<false, ...>); bytes = FETCH_SIZE*1024*2 + WRITE_SIZE*1024 (gfx950 x2 read-side "
                   "correction); memory-side counter, Infinity-Cache hits included" % config}


def roofline_report(config, spp, modified, alg_bytes_per_launch, avg_ms, scene_bytes, extra, live=None):
    """The dominant kernel against the memory roofline.  `achieved` is the ALGORITHMIC rate (bytes the
    traversal must fetch per launch / launch time); when the scene fits the 256 MiB Infinity Cache the bound is not HBM and the
    algorithmic rate is not comparable with the HBM peak (L2 / Infinity-Cache hits serve part of it), so `frac` is then taken
    from what actually crossed the L2's memory side (min(algorithmic, counter traffic)) -- or is null without counters."""
    alg_gbps = alg_bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    resident = scene_bytes <= INFINITY_CACHE_BYTES
    ent, reason = pmc_entry(config, spp, modified)
    r = {"kernel": "k_trace_nearest", "unit": "GB/s", "peak": HBM_PEAK_GBPS,
         # `bound` names the ROOF the kernel is priced against (the contract's "hbm" | "mfma": a gather, no matrix work); `limited_by` says what the
         # counters of this build show to be the tightest limit in fact
         "bound": "hbm", "limited_by": "l2-miss/fabric gather (Infinity-Cache resident)" if resident else "hbm",
         "scene_bytes": int(scene_bytes), "alg_gbps": round(alg_gbps, 1),
         "alg_frac_of_hbm_peak": round(alg_gbps / HBM_PEAK_GBPS, 4),
         "alg_bytes_per_launch": round(alg_bytes_per_launch), "node_bytes_per_visit": NODE_BYTES_FETCHED,
         "alg_formula": "48 B fetched per inner-node visit (3 x dwordx4 of a 64-B-stride node; SURVEY 8d assumed 128-B nodes) + 48 B per triangle test + 48 B per ray (32-B ray read, 16-B hit write)",
         "avg_launch_ms": round(avg_ms, 4)}
    r.update(extra)
    live_ok = bool(live) and "error" not in live
    if live and not live_ok:
        r["traffic_live_error"] = live["error"]
    if ent is None and live_ok:
        ent = {"hbm_bytes_per_launch": live["hbm_bytes_per_launch"]}          # no committed passes of this build: the live bytes alone (no SQ ceilings, no hit rate)
        r["committed_passes"] = reason
    if ent is not None:
        traffic = float(live["hbm_bytes_per_launch"] if live_ok else ent["hbm_bytes_per_launch"])
        t_gbps = traffic / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        hit, miss = ent.get("tcc_hit_per_launch"), ent.get("tcc_miss_per_launch")
        if live_ok:
            measured = (f"IN THIS RUN: two counter passes ({live['seconds']} s) started by this process after its timed region, {live['launches_per_pass']} launches each; "
                        "the SQ / TCC figures of roofline.ceilings and l2_hit_rate are the committed passes of the same build (profiles/pmc_traffic.json)")
            r["traffic_this_run"] = {k: live[k] for k in ("fetch_size_kb_per_launch", "write_size_kb_per_launch", "launches_per_pass", "seconds")}
            if "source_hash" in ent:
                r["traffic_committed"] = float(ent["hbm_bytes_per_launch"])
                r["traffic_this_run_over_committed"] = round(traffic / max(float(ent["hbm_bytes_per_launch"]), 1.0), 4)
        else:
            measured = (f"NOT in this run: committed counter passes (profiles/pmc_traffic.json, collected by profiles/pmc_collect.sh on kernel build "
                        f"{ent.get('source_hash')} = the sources of this library, hash re-checked at run time); this run supplies avg_launch_ms only, so "
                        "traffic_gbps / frac = committed bytes per launch / THIS run's launch time")
        r.update({"traffic": traffic, "traffic_source": live["how"] if live_ok else ent.get("how", "profiles/pmc_traffic.json"),
                  "traffic_measured": measured,
                  "traffic_gbps": round(t_gbps, 1), "traffic_frac_of_peak": round(t_gbps / HBM_PEAK_GBPS, 4),
                  "traffic_frac_of_achievable": round(t_gbps / HBM_ACHIEVABLE_GBPS, 4),
                  "traffic_over_alg": round(traffic / max(alg_bytes_per_launch, 1.0), 3),
                  "l2_hit_rate": round(hit / (hit + miss), 3) if hit and miss else None,
                  "pmc_avg_launch_ms": ent.get("avg_launch_ms")})
        ceil = roofline_ceilings(dict(ent, hbm_bytes_per_launch=traffic), avg_ms) if "source_hash" in ent else None
        if ceil:
            r["ceilings"] = ceil
            if ceil.get("binding"):
                names = {"hbm": "hbm (memory-side traffic)", "l2": "l2 request rate", "valu_issue": "VALU issue"}
                r["limited_by"] = f"{names[ceil['binding']]}: {ceil[ceil['binding']]:.2f} of its ceiling (measured: roofline.ceilings)"
        elif resident and t_gbps < 0.75 * HBM_ACHIEVABLE_GBPS:
            r["limited_by"] = "memory side not saturated (Infinity-Cache resident); no SQ counters for this build"
        if resident or t_gbps < alg_gbps:
            r["achieved"] = round(min(alg_gbps, t_gbps), 1)
            r["achieved_basis"] = "min(algorithmic, memory-side counter traffic): bytes that were both needed and crossed the L2's memory side"
            r["frac_note"] = (("cache-resident scene" if resident else "the caches serve part of the gather (traffic_over_alg < 1: samples of a pixel travel together and share most of their walk)") +
                              ": frac counts only what crossed the L2s' memory side, so it FALLS when the L2s serve more of the gather "
                              "(the kernel gets faster): the rate against the algorithmic bytes is alg_frac_of_hbm_peak" + ("; the HBM-resident config is C5" if resident else ""))
        else:
            r["achieved"] = round(alg_gbps, 1)
            r["achieved_basis"] = "algorithmic bytes / kernel time (traffic_over_alg > 1 = over-fetch)"
        r["frac"] = round(r["achieved"] / HBM_PEAK_GBPS, 4)
        # the same counter passes of THIS build for the other single-GPU configs (profiles/pmc_traffic.json), so that the line of the cache-resident
        # headline config also shows the HBM-resident one: C5 (10 M triangles, 0.97 GB of nodes + triangles) is where the path meets the memory roof
        others = {}
        try:
            pmc = json.load(open(PMC_FILE))
            for cfg, e in (pmc.get("configs") or {}).items():
                if cfg != config and e.get("source_hash") == kernel_source_hash() and e.get("avg_launch_ms"):
                    c2 = roofline_ceilings(e, e["avg_launch_ms"]) or {}
                    others[cfg] = {"workload": e.get("workload"), "avg_launch_ms": round(e["avg_launch_ms"], 3),
                                   "hbm_frac_memory_side": c2.get("hbm"), "valu_issue": c2.get("valu_issue"), "l2": c2.get("l2"), "binding": c2.get("binding")}
        except (OSError, ValueError, KeyError):
            pass
        if others:
            r["other_configs_same_build"] = others
    else:
        r.update({"traffic": None, "traffic_reason": reason, "l2_hit_rate": None})
        r["achieved"] = round(alg_gbps, 1)
        r["achieved_basis"] = "algorithmic bytes / kernel time"
        if resident:
            r["frac"] = None
            r["frac_reason"] = "scene is Infinity-Cache resident and there is no counter traffic for this build: the algorithmic rate is not comparable with the HBM peak"
        else:
            r["frac"] = round(alg_gbps / HBM_PEAK_GBPS, 4)
    return r


L2_PEAK_GBPS = 34500.0            # /opt/skills/guides/MI355X_MICROARCH.md "L2 (per XCD)": ~34.5 TB/s aggregate
N_SIMDS = 256 * 4                 # 256 CUs x 4 SIMDs


def roofline_ceilings(ent, avg_ms):
    """Which roof the dominant kernel is under, from the committed SQ / TCC counter passes of THIS build (profiles/pmc_collect.sh).  Units as
    calibrated on gfx950 (profiles/r3/counter_units.md): GRBM_GUI_ACTIVE is summed over the 8 XCDs (/ 8 = shader cycles of the launch, 2.34 GHz);
    SQ_ACTIVE_INST_* and SQ_WAVE_CYCLES are in quad-cycles summed over all waves; a wave64 VALU instruction occupies its 16-lane SIMD for 4 cycles.
      hbm         memory-side bytes / time against the 8 TB/s peak
      l2          requests that reached the L2s x 128-B line / time against the ~34.5 TB/s the guide measures for the L2s -- an UPPER estimate of
                  the L2-side load (a divergent 16-B lane request occupies a line slot but moves less)
      valu_issue  4 x SQ_ACTIVE_INST_VALU / (1024 SIMDs x cycles): the share of all SIMD issue cycles of the launch spent issuing VALU instructions
      lane_util   SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU): active lanes per issued VALU instruction -- a multiplier on what the issue
                  slots achieve, not a ceiling of its own
    `binding` = the largest of hbm / l2 / valu_issue."""
    c = ent.get("counters_per_launch") or {}
    t = avg_ms * 1e-3
    if t <= 0:
        return None
    out = {"hbm": round(float(ent["hbm_bytes_per_launch"]) / t / 1e9 / HBM_PEAK_GBPS, 4)}
    if c.get("TCC_REQ_sum"):
        out["l2"] = round(c["TCC_REQ_sum"] * 128.0 / t / 1e9 / L2_PEAK_GBPS, 4)
        out["l2_requests_per_launch"] = c["TCC_REQ_sum"]
    cycles = c["GRBM_GUI_ACTIVE"] / 8.0 if c.get("GRBM_GUI_ACTIVE") else None
    pmc_ms = ent.get("avg_launch_ms") or avg_ms                      # the counters belong to the profiled launches
    if cycles is None:
        cycles = pmc_ms * 1e-3 * 2.34e9
    else:
        out["shader_clock_ghz"] = round(cycles / (pmc_ms * 1e-3) / 1e9, 3)
    if c.get("SQ_ACTIVE_INST_VALU") and c.get("SQ_WAVE_CYCLES") and c.get("SQ_WAVES"):
        # same-pass, clock-free form: the persistent waves live for the whole launch, SQ_WAVES / 1024 of them share a SIMD, so a SIMD's time is
        # SQ_WAVE_CYCLES / (waves per SIMD) and its VALU is issuing for SQ_ACTIVE_INST_VALU of it (both quad-cycles).  Comes out at 1.00-1.05 on the
        # cache-resident configs (the waves' ramp-up / drain is not in SQ_WAVE_CYCLES): saturated.  The clock-based form 4 x ACTIVE / (1024 x cycles)
        # with cycles from another pass agrees within the pass-to-pass spread of the launch time.
        out["valu_issue"] = round(c["SQ_ACTIVE_INST_VALU"] * (c["SQ_WAVES"] / float(N_SIMDS)) / c["SQ_WAVE_CYCLES"], 4)
        out["valu_issue_clock_based"] = round(4.0 * c["SQ_ACTIVE_INST_VALU"] / (N_SIMDS * cycles), 4)
    elif c.get("SQ_ACTIVE_INST_VALU"):
        out["valu_issue"] = round(4.0 * c["SQ_ACTIVE_INST_VALU"] / (N_SIMDS * cycles), 4)
    if c.get("SQ_THREAD_CYCLES_VALU") and c.get("SQ_ACTIVE_INST_VALU"):
        out["lane_util"] = round(c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"]), 4)
    if c.get("SQ_INSTS_VALU"):
        out["valu_insts_per_launch"] = c["SQ_INSTS_VALU"]
        out["valu_issue_floor_ms"] = round(c["SQ_INSTS_VALU"] * 4.0 / N_SIMDS / (cycles / (pmc_ms * 1e-3)) * 1e3, 3)     # every instruction issued back to back
    if c.get("SQ_WAVE_CYCLES") and c.get("SQ_WAIT_ANY") is not None:
        out["wave_time_split"] = {k: round(c[n] / c["SQ_WAVE_CYCLES"], 3) for k, n in (("waiting", "SQ_WAIT_ANY"), ("issue_stalled", "SQ_WAIT_INST_ANY"), ("issuing", "SQ_ACTIVE_INST_ANY")) if c.get(n) is not None}
    if c.get("SQ_INSTS_VMEM_RD"):
        out["vmem_rd_insts_per_launch"] = c["SQ_INSTS_VMEM_RD"]
    cands = {k: out[k] for k in ("hbm", "l2", "valu_issue") if k in out}
    if cands:
        out["binding"] = max(cands, key=cands.get)
    return out


def free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def visible_gpu_count():
    """GPUs this process could use, WITHOUT touching the HIP runtime or importing torch (the parent of spawned ranks never does): the KFD topology of
    the ROCm driver, cut by the *_VISIBLE_DEVICES lists.  None when it cannot be told from here (the ranks then check for themselves after importing torch)."""
    import re
    if not os.path.exists("/dev/kfd"):
        return 0                                             # no ROCm device node at all
    n = None
    try:
        base = "/sys/class/kfd/kfd/topology/nodes"
        n = 0
        for d in os.listdir(base):
            m = re.search(r"^simd_count\s+(\d+)", open(os.path.join(base, d, "properties")).read(), re.M)
            if m and int(m.group(1)) > 0:
                n += 1
    except OSError:
        return None
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip()]))
    return n


def refuse_without_enough_gpus(wanted, have):
    sys.exit(f"bench.py: --gpus {wanted} needs {wanted} GPUs on this node, {have} visible -- nothing was started "
             "(CRH_BENCH_SHARE_DEVICE=1 CRH_BENCH_BACKEND=gloo rehearses the N-rank flow on one GPU)")


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` outside a launcher: start N fresh rank processes (this parent has not imported torch nor
    touched the HIP runtime, and never does) and wait for them.  Rank 0 prints the JSON line on the inherited stdout."""
    port = os.environ.get("MASTER_PORT") or str(free_port())
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=port, CRH_BENCH_SPAWNED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    rc = 0
    deadline = None
    while procs:
        for p in list(procs):
            code = p.poll()
            if code is None:
                continue
            procs.remove(p)
            if code != 0 and rc == 0:
                rc = code
                deadline = time.time() + 30.0          # a rank died: the others would wait in a collective forever
        if deadline is not None and time.time() > deadline:
            for p in procs:
                p.kill()
        time.sleep(0.05)
    return rc


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="C3", choices=["C1", "C2", "C3", "C4", "C5", "CAD1M"])
    ap.add_argument("--scaling", default=None, choices=["weak", "strong"])
    ap.add_argument("--spp", type=int, default=0, help="samples per pixel per step (one crh_render_tiles call); 0 = 512 for C3, 256 for C5 and C2, 1024 for C1, 4096 for C4")
    ap.add_argument("--tris", type=int, default=0, help="override triangle count (debug)")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-parity", action="store_true", help="skip the parity gates (the oracle re-renders sampled tiles of the TIMED frames)")
    ap.add_argument("--parity-seconds", type=float, default=8.0, help="CPU budget of the whole-timed-region parity gate")
    ap.add_argument("--step0-seconds", type=float, default=6.0, help="CPU budget of the first-timed-step parity gate (at least 32 tiles whatever it costs)")
    ap.add_argument("--no-interactive", action="store_true", help="skip the 1-spp-per-Redraw figure")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--other-configs", default=None,
                    help="comma list of further single-GPU configs timed AFTER the headline as short legs, each with its own parity gates and roofline "
                         "(config.other_configs_timed); default C5,C2,C1,CAD1M for the default headline run at N = 1, none otherwise; 'none' switches them off")
    ap.add_argument("--other-steps", type=int, default=4)
    ap.add_argument("--assemble", default="auto", choices=["auto", "reduce", "gather"],
                    help="N > 1 exchange step: full-frame RCCL reduce, gather of owned tiles, or whichever is faster on this fabric (timed before the run)")
    ap.add_argument("--live-traffic", default=None, choices=["on", "off"],
                    help="roofline.traffic from counter passes taken IN this run: after the timed region the same command runs twice more as a child under "
                         "`rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (2 steps each, about 20 s per pass); default on for the plain headline run at N = 1, "
                         "off otherwise; without rocprofv3, or when this process is itself being profiled, the committed passes (profiles/pmc_traffic.json) are used")
    ap.add_argument("--allow-unchecked", action="store_true",
                    help="exit 0 even when NO parity gate of the headline leg produced a verdict (checker could not run); default: exit code 3 (a failed gate: 1)")
    ap.add_argument("--no-shared-build", action="store_true", help="N > 1: every rank builds its own BVH (default: rank 0 builds, the others take its tree)")
    args = ap.parse_args(argv)
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if args.scaling is None:
        args.scaling = "strong" if args.config == "C4" else "weak"
    if args.steps is None:
        args.steps = 1 if args.config == "C4" else 4
    if args.other_configs is None:
        plain = args.gpus == 1 and args.config == "C3" and not (args.tris or args.width or args.height or args.spp)
        args.other_configs = "C5,C2,C1,CAD1M" if plain else "none"
    if args.live_traffic is None:
        plain = args.gpus == 1 and args.config == "C3" and not (args.tris or args.width or args.height or args.spp) and args.other_configs != "none"
        args.live_traffic = "on" if plain else "off"
    args.other_list = [c for c in args.other_configs.split(",") if c and c != "none"]
    for c in args.other_list:
        if c not in ("C1", "C2", "C3", "C5", "CAD1M"):
            ap.error(f"--other-configs: unknown config {c}")
    return args


def main():
    args = parse_args()
    rehearsal = os.environ.get("CRH_BENCH_SHARE_DEVICE") == "1" or os.environ.get("CRH_BENCH_RANK_PROBE") == "1"
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        have = None if rehearsal else visible_gpu_count()
        if have is not None and have < args.gpus:
            refuse_without_enough_gpus(args.gpus, have)         # one line, at once: not N ranks waiting for each other in a rendezvous (round-5 verdict, item 6)
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))          # nothing GPU-related has been imported yet

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")

    if os.environ.get("CRH_BENCH_RANK_PROBE") == "1":
        # launcher self-test (tests/test_bench_cli.py, CPU): rendezvous over gloo, one all-reduce, rank 0 reports what it saw
        import torch
        import torch.distributed as dist
        dist.init_process_group("gloo")
        t = torch.ones(1)
        dist.all_reduce(t)
        if rank == 0:
            print(json.dumps({"probe": True, "n_gpus": world, "rccl_ranks": dist.get_world_size(), "sum": int(t.item()),
                              "spawned": os.environ.get("CRH_BENCH_SPAWNED") == "1"}), flush=True)
        dist.barrier()
        dist.destroy_process_group()
        return

    # before anything initialises HIP: free-running Redraw()s keep eight frames in flight, one stream each, and the runtime maps streams onto
    # GPU_MAX_HW_QUEUES hardware queues (default 4).  bench.py is the HOST here and exports the variable itself -- the library never touches the
    # environment (crh_query_pipeline_capacity reports what it found); only the `interactive` figures depend on it
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    import torch
    dist = None
    backend = None
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("CRH_BENCH_BACKEND", "nccl")     # "gloo" + CRH_BENCH_SHARE_DEVICE=1: rehearsal of the N > 1 flow on one GPU
        if os.environ.get("CRH_BENCH_SHARE_DEVICE") == "1":
            local = 0
        elif torch.cuda.device_count() < int(os.environ.get("LOCAL_WORLD_SIZE", str(world))):
            # under a launcher (torchrun) nobody checked: EVERY rank sees the same count and leaves before the rendezvous, so none waits for another
            refuse_without_enough_gpus(int(os.environ.get("LOCAL_WORLD_SIZE", str(world))), torch.cuda.device_count())
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
        else:
            dist.init_process_group(backend)

    # N ranks on one host share its CPUs.  With the shared build (default) rank 0 builds the tree with every thread and the others wait; without it every
    # rank builds the same BVH with its share of the threads (8 ranks x all threads oversubscribed the 16 usable CPUs of a GPU box 8-fold)
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    if local_world > 1 and args.no_shared_build and not os.environ.get("CRH_BUILD_THREADS"):
        os.environ["CRH_BUILD_THREADS"] = str(max(1, usable_cpus() // local_world))
    if local_world > 1 and not args.no_shared_build:
        os.environ.pop("LOCAL_WORLD_SIZE", None)                  # the builder would take 1 / LOCAL_WORLD_SIZE of the CPUs (bvh_builder.cpp): one rank builds here

    ctxt = {"world": world, "rank": rank, "local": local, "dist": dist, "backend": backend, "torch": torch}
    out, failed, errors = run_leg(args, ctxt, args.config, args.steps, args.warmup, args.spp, headline=True)

    # ---- the other single-GPU configs of BASELINE.json, driver-timed in the same run (verdict r3 item 1): short legs, own parity gates, own roofline.
    # `value` stays the headline config; C5 (10 M triangles, 4K) is the one whose scene does not fit the caches, i.e. where HBM is the roof.
    if rank == 0 and world == 1 and args.other_list:
        others = {}
        for cfg in args.other_list:
            t0 = time.time()
            try:
                o, f, e = run_leg(args, ctxt, cfg, args.other_steps, 1, 0, headline=False)
                others[cfg] = compact_leg(o)
                others[cfg]["leg_wall_s"] = round(time.time() - t0, 1)
                failed = failed or f
                errors += e
            except Exception as e:                              # a leg that cannot run must not lose the headline
                others[cfg] = {"error": f"{type(e).__name__}: {e}"}
                errors.append(f"{cfg}: {type(e).__name__}: {e}")
        out["config"]["other_configs_timed"] = others
        out["config"]["other_configs_note"] = ("timed in this very run after the headline, inputs resident in HBM, 1 warm-up + %d timed steps each, same kernels and schedule; "
                                               "NOT part of `value`" % args.other_steps)
    unchecked = False
    if rank == 0:
        if errors:
            out["checker_errors"] = errors
        # a headline number no gate looked at is not a measurement (a parity gate accompanies every number)
        verdicts = [p for p in (out.get("parity"), out.get("parity_step0")) if p is not None and "error" not in p]
        unchecked = not args.no_parity and not verdicts
        flatten_line(out)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()                                       # rank 0 may still be in its counting pass
        dist.destroy_process_group()
    if failed:
        sys.exit("bench.py: PARITY GATE FAILED -- timed frames differ from the oracle (see \"parity\" / \"parity_step0\" in the JSON line)")
    if errors:
        print("bench.py: a checker leg could not run (see \"checker_errors\" in the JSON line): " + "; ".join(errors), file=sys.stderr)
    if unchecked and not args.allow_unchecked:
        print("bench.py: NO parity gate of the headline leg produced a verdict -- the number is unchecked (exit 3; --allow-unchecked accepts it)", file=sys.stderr)
        sys.exit(3)


def flatten_line(out):
    """The driver's record keeps top-level scalars and the scalars inside `config` / `roofline` (nested objects are dropped, the stdout tail starts somewhere in the
    line): every leg's rate, gate and fractions, the interactive figures and the measured ceilings are therefore ALSO written as flat scalar keys -- at the top level,
    inside `config` and `roofline` -- and once more as one small object, `legs_summary`, the LAST key of the line (verdict r4 item 3)."""
    cfg, roof = out.get("config") or {}, out.get("roofline") or {}
    flat, wide = {}, {}          # flat: top level + config + legs_summary (the tail); wide: top level + config only
    def gate(o):
        ps = [p for p in (o.get("parity"), o.get("parity_step0")) if p is not None]
        return None if not ps or any("error" in p for p in ps) else all(bool(p.get("bit_exact")) for p in ps)
    def leg(tag, o, r):
        flat[f"{tag}_mrays"] = o.get("value"); flat[f"{tag}_ms_per_step"] = o.get("ms_per_step")
        flat[f"{tag}_frac_counter"] = r.get("traffic_frac", r.get("traffic_frac_of_peak")); flat[f"{tag}_alg_frac"] = r.get("alg_frac", r.get("alg_frac_of_hbm_peak"))
        flat[f"{tag}_parity_bit_exact"] = gate(o)
        c = r.get("ceilings") or {}
        flat[f"{tag}_valu_issue"] = c.get("valu_issue"); flat[f"{tag}_lane_util"] = c.get("lane_util")
        # round 6: what the CAD-like leg is there to show -- kept out of the 2000-character tail for the soup legs (`wide` = flat keys only)
        ff = o.get("first_frame_after_a_restart_ms", ((o.get("config") or {}).get("interactive") or {}).get("first_frame_after_a_restart_ms"))
        for k, val in (("nodes_per_ray", r.get("nodes_per_ray")), ("tris_per_ray", r.get("tris_per_ray")), ("packet_fallback_fraction", r.get("packet_fallback_fraction")),
                       ("first_frame_ms", ff if tag != head else None)):
            if val is not None:
                (flat if tag.startswith("cad") else wide)[f"{tag}_{k}"] = val
    head = (cfg.get("workload") or "C?").split(":")[0].lower()
    leg(head, out, roof)
    for name, o in (cfg.get("other_configs_timed") or {}).items():
        if "error" in o:
            flat[f"{name.lower()}_error"] = o["error"][:120]
        else:
            leg(name.lower(), o, o.get("roofline") or {})
    it = cfg.get("interactive") or {}
    for k_src, k_dst in (("redraw_per_s_lookahead_1", "interactive_redraw_per_s"), ("first_frame_after_a_restart_ms", "interactive_first_frame_ms"),
                         ("drag_frames_per_s", "interactive_drag_frames_per_s"), ("displayed_frames_per_s", "interactive_displayed_frames_per_s"),
                         ("redraw_per_s_lookahead_64", "interactive_redraw_per_s_lookahead_64"),
                         ("displayed_grays_per_s", "interactive_displayed_grays_per_s"), ("first_frame_grays_per_s", "interactive_first_frame_grays_per_s")):
        if k_src in it:
            flat[k_dst] = it[k_src]
    if "drag_grays_per_s" in it:
        wide["interactive_drag_grays_per_s"] = it["drag_grays_per_s"]
    g = it.get("gates") or {}
    if g:
        flat["interactive_gates_bit_exact"] = None if "error" in g else bool(g.get("drag_last_frame_bit_exact") and g.get("displayed_frame_bit_exact"))
    # the only TRUE HBM fraction of the run: C5's scene (0.97 GB) does not fit the caches, so its counter fraction is the HBM-resident figure; the headline
    # config's own `frac` is that of a cache-resident scene (round-5 verdict, item 8)
    c5 = ((cfg.get("other_configs_timed") or {}).get("C5") or {}).get("roofline") or {}
    if c5.get("traffic_frac") is not None:
        roof["hbm_resident_frac"] = c5["traffic_frac"]
        roof["hbm_resident_frac_note"] = "C5 leg of this run (10 M triangles, 4K): memory-side counter traffic / 8 TB/s, leg average over its traversal launches"
    ceil = roof.get("ceilings") or {}
    for k in ("valu_issue", "lane_util", "hbm", "l2"):
        if k in ceil and isinstance(ceil[k], (int, float)):
            roof[k if k in ("valu_issue", "lane_util") else k + "_frac_ceiling"] = ceil[k]
    cfg.update(wide); cfg.update(flat)
    out.pop("legs_summary", None)
    out.update(wide); out.update(flat)
    out["legs_summary"] = flat          # last key: the tail of the line carries every leg


def compact_leg(o):
    """What config.other_configs_timed keeps of a leg's full line."""
    r = o.get("roofline") or {}
    ceil = r.get("ceilings") or {}
    keep = lambda p: None if p is None else {k: p.get(k) for k in ("bit_exact", "rel_l2", "tiles", "pixels", "spp", "first_sample", "schedule", "error") if k in p}
    return {"workload": o["config"]["workload"], "value": o["value"], "unit": o["unit"], "ms_per_step": o["ms_per_step"], "steps": o["steps"], "warmup": o["warmup"],
            "spp_per_step": o["config"]["spp_per_step_per_rank"], "msamples_per_s": o["config"]["msamples_per_s"], "build_upload_s": o["config"]["build_upload_s"],
            "parity": keep(o.get("parity")), "parity_step0": keep(o.get("parity_step0")),
            "first_frame_after_a_restart_ms": (o["config"].get("interactive") or {}).get("first_frame_after_a_restart_ms"),
            "roofline": {"kernel": r.get("kernel"), "avg_launch_ms": r.get("avg_launch_ms"), "launches": r.get("launches"), "kernel_time_share": r.get("kernel_time_share"),
                         "scene_bytes": r.get("scene_bytes"), "alg_gbps": r.get("alg_gbps"), "alg_frac": r.get("alg_frac_of_hbm_peak"),
                         "traffic_gbps": r.get("traffic_gbps"), "traffic_frac": r.get("traffic_frac_of_peak"), "traffic_measured": r.get("traffic_measured"),
                         "traffic": r.get("traffic"), "traffic_this_run_over_committed": r.get("traffic_this_run_over_committed"), "traffic_live_error": r.get("traffic_live_error"),
                         "traffic_reason": r.get("traffic_reason"), "frac": r.get("frac"), "achieved_basis": r.get("achieved_basis"),
                         "nodes_per_ray": r.get("nodes_per_ray"), "tris_per_ray": r.get("tris_per_ray"),
                         "packet_rays": r.get("packet_rays"), "packet_fallback_rays": r.get("packet_fallback_rays"), "packet_fallback_fraction": r.get("packet_fallback_fraction"),
                         "ceilings": {k: ceil.get(k) for k in ("hbm", "l2", "valu_issue", "lane_util", "binding") if k in ceil} or None,
                         "limited_by": r.get("limited_by")}}


def run_leg(args, ctxt, config, steps, warmup, spp_arg, headline):
    """One workload, measured: scene hand-over, warm-up, EXACTLY `steps` timed steps between barriers, counting pass, parity gates, roofline.
    Returns (json dict on rank 0 | None, parity_failed, checker_errors)."""
    import numpy as np
    world, rank, local, dist, backend, torch = (ctxt[k] for k in ("world", "rank", "local", "dist", "backend", "torch"))
    from cadrays_amd import scenes, sharding
    from cadrays_amd.view import View
    n_gpus = world
    scaling = args.scaling if headline else "weak"
    errors = []

    scene_cfg = "C3" if config == "C4" else config
    ov = headline                                            # command-line overrides of the workload apply to the headline leg only
    t0 = time.time()
    sc = scenes.baseline_config(scene_cfg, (args.width or None) if ov else None, (args.height or None) if ov else None, (args.tris or None) if ov else None)
    gen_s = time.time() - t0
    t0 = time.time()
    v = View(local)
    dev = torch.device(f"cuda:{local}") if backend == "nccl" else torch.device("cpu")
    if world > 1 and not args.no_shared_build:
        share = sharding.load_scene_shared(v, sc, dist, dev)
    else:
        v.load_scene(sc); share = None
    build_s = time.time() - t0
    fb = sharding.DeviceFramebuffer(v) if world > 1 else None
    tiles = sharding.tiles_for_rank(v.n_tiles(), rank, world, sharding.tiles_x_of(v))       # Morton-interleaved across the ranks
    spp = spp_arg
    if spp <= 0:
        # samples per pixel of one step = one crh_render_tiles call.  The library cuts a call into batches of <= 2^29 paths, tile groups first (up to 1024
        # samples of a pixel travel together: crh_schedule.cpp), so a step is given enough samples for that to matter: 512 at 1080p (C3: two batches of
        # 1024 tiles), 256 at 4K (C5: four batches of 2048 tiles), C4 its named 4096; C2 its named 256 (one batch of 2040 tiles; it does not care: 128 / 256 / 512 give 5572 / 5576 / 5585), C1 1024 as in round 3.  A workload
        # overridden on the command line gets what fills 2^28 slots, in multiples of 64.
        named = {"C3": 512, "C4": 4096, "C5": 256, "C2": 256, "C1": 1024, "CAD1M": 512}
        if config in named and not (ov and (args.tris or args.width or args.height)):
            spp = named[config]
        else:
            spp = max(1, (256 << 20) // (v.n_tiles() * sc.params.tile_size ** 2))
            if spp >= 64:
                spp &= ~63
    spp_step = spp * world if scaling == "weak" else spp      # samples per pixel each rank renders per step

    def barrier():
        v.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    # ---- N > 1: the exchange step.  Full-frame reduce (sum of disjoint supports) or gather of owned tiles (1 / N of the bytes): timed on THIS fabric, the
    # faster one is used in the timed steps (--assemble forces one); both assemble the same bits
    assembled = [None]
    assemble_info = None
    gather = sharding.TileGather(v.width, v.height, v.tile_size, world, fb.tensor.device) if world > 1 else None

    def assemble_reduce():
        return sharding.reduce_framebuffer(fb.tensor, 0)

    def assemble_gather():
        return gather.assemble(fb.tensor, rank, 0)

    assemble = assemble_reduce
    rccl_ranks = 1
    if dist is not None:                                      # RCCL builds its rings / channels on first use: keep that out of the timed steps
        timing = {}
        for name, fn in (("reduce", assemble_reduce), ("gather", assemble_gather)):
            try:
                fn(); barrier()
                t1 = time.perf_counter()
                for _ in range(3):
                    fn()
                barrier()
                tt = torch.tensor([(time.perf_counter() - t1) / 3 * 1e3], dtype=torch.float64, device=dev)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                timing[name] = round(float(tt.item()), 3)
            except (RuntimeError, NotImplementedError) as e:          # a backend without this collective refuses it on every rank alike, before anything is sent
                if name == "reduce":
                    raise
                timing[name] = None
                timing[name + "_error"] = f"{type(e).__name__}: {e}"[:200]
        if args.assemble == "gather" and timing["gather"] is None:
            sys.exit("bench.py: --assemble gather, but this backend refused the gather: " + timing["gather_error"])
        mode = args.assemble if args.assemble != "auto" else ("gather" if timing["gather"] is not None and timing["gather"] < timing["reduce"] else "reduce")
        assemble = assemble_gather if mode == "gather" else assemble_reduce
        assemble_info = {"mode": mode, "chosen_by": "--assemble" if args.assemble != "auto" else "measurement before the timed steps (3 calls each, max over ranks)",
                         "reduce_ms": timing["reduce"], "gather_ms": timing["gather"],
                         "reduce_bytes_per_rank": int(v.width * v.height * 16), "gather_bytes_per_rank": int(gather.bytes_per_rank)}
        if timing.get("gather_error"):
            assemble_info["gather_error"] = timing["gather_error"]
        rccl_ranks = dist.get_world_size()                    # as seen after the first collectives

    t_render = [0.0]; t_assemble = [0.0]

    def step(i):
        if dist is None:
            v.render_tiles(tiles, i * spp_step, spp_step)
            return
        t1 = time.perf_counter()
        v.render_tiles(tiles, i * spp_step, spp_step)
        v.sync()                                          # the accumulator is written on the context's own stream
        t2 = time.perf_counter()
        assembled[0] = assemble()                         # returns synchronised
        t_render[0] += t2 - t1; t_assemble[0] += time.perf_counter() - t2

    for i in range(warmup):
        step(i)
    barrier()
    v.reset()
    v.enable_kernel_timing(True)
    t_render[0] = t_assemble[0] = 0.0
    barrier()
    t0 = time.perf_counter()
    for i in range(steps):
        step(warmup + i)
    barrier()
    dt = time.perf_counter() - t0
    v.enable_kernel_timing(False)
    kt = v.kernel_timing()
    st = v.stats()
    pk = v.packet_stats()                                     # camera rays of the timed steps walked as packets / handed to the per-ray fall-back pass
    # what the TIMED steps themselves accumulated (rank 0, N = 1): the parity gate below compares exactly these pixels with the oracle
    timed_hdr = v.read_hdr() if (rank == 0 and world == 1 and not args.no_parity) else None
    if rank == 0 and world > 1 and not args.no_parity and assembled[0] is not None:
        # N > 1: the frame the LAST timed step's exchange assembled on rank 0 (every rank's tiles, the same samples) goes through the same gate
        timed_hdr = assembled[0][..., :3].contiguous().cpu().numpy()
    timed_first, timed_n = warmup * spp_step, steps * spp_step
    per_rank = None
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        cnt = torch.tensor([st["rays_nearest"], st["rays_any"], st["samples"]], dtype=torch.float64, device=dev)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        rays_n, rays_a, samples = (float(x) for x in cnt.tolist())
        mine = torch.tensor([t_render[0] * 1e3 / steps, t_assemble[0] * 1e3 / steps, float(st["rays_nearest"] + st["rays_any"]), float(len(tiles)),
                             (share or {}).get("seconds", {}).get("load_scene", 0.0) + (share or {}).get("seconds", {}).get("load_prebuilt", 0.0), build_s], dtype=torch.float64, device=dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [{"rank": r, "ms_render": round(float(a[0]), 3), "ms_reduce": round(float(a[1]), 3), "rays": int(a[2]), "tiles": int(a[3]),
                     "scene_hand_over_s": round(float(a[5]), 2)} for r, a in enumerate(x.tolist() for x in allr)]
    else:
        rays_n, rays_a, samples = float(st["rays_nearest"]), float(st["rays_any"]), float(st["samples"])

    # ---- counting pass (untimed): the same frames again with node / triangle counters on (rank 0's shard)
    roof = None
    wide_batch = (spp_step * len(tiles) * sc.params.tile_size ** 2) > (12 << 20) and (spp_step & -spp_step) >= 64      # crh_context.h: packets from 64 consecutive samples per pixel on
    if rank == 0:
        v.reset(); v.enable_counters(True)
        for i in range(steps):
            v.render_tiles(tiles, (warmup + i) * spp_step, spp_step)
        cs = v.stats()
        v.enable_counters(False)
        assert cs["rays_nearest"] == st["rays_nearest"], "counting pass traced different rays"
        launches = max(kt["trace_nearest_launches"], 1)
        alg_bytes = float(NODE_BYTES_FETCHED) * cs["nodes_nearest"] + 48.0 * cs["tris_nearest"] + 48.0 * cs["rays_nearest"]
        avg_ms = kt["trace_nearest_ms_total"] / launches
        mem = v.scene_bytes()
        modified = bool(ov and (args.tris or args.width or args.height))
        # the memory-side counters of THIS run (two child passes under rocprofv3, after the timed region; the parent idles meanwhile)
        live = None
        if world == 1 and args.live_traffic == "on" and not modified:          # the headline and every other-config leg
            budget_was = v.get_path_budget()                                    # restored as it was (the library's default or the caller's), not as a constant
            v.set_path_budget(1 << 20)                                          # the child holds a full path budget of its own (up to 105 GB): this process's goes first
            live = live_traffic(config)
            v.set_path_budget(budget_was)
        roof = roofline_report(config if world == 1 else f"{config}@{world}", spp_step, modified,
                               alg_bytes / launches, avg_ms, mem["nodes"] + mem["triangles"], live=live, extra={
            "launches": int(launches),
            "kernel_time_share": round(kt["trace_nearest_ms_total"] / max(kt["render_ms_total"], 1e-9), 3),
            "nodes_per_ray": round(cs["nodes_nearest"] / max(cs["rays_nearest"], 1), 2),
            "tris_per_ray": round(cs["tris_nearest"] / max(cs["rays_nearest"], 1), 2),
            "packet_rays": pk["packet_rays"], "packet_fallback_rays": pk["fallback_rays"],
            "packet_fallback_fraction": round(pk["fallback_rays"] / pk["packet_rays"], 6) if pk["packet_rays"] else None,
            "camera_rays": ("bounce 0 of a wide batch is walked by k_trace_packets (one packet per wavefront: 64 samples of a pixel share the node fetches) + the per-ray "
                            "fall-back pass for rays that met two triangles at exactly the same distance; it is one of the `launches`, timed like the others; "
                            "nodes_per_ray / tris_per_ray / alg_bytes are those of the spec's per-ray walk (counting pass), which the packet walk does not exceed per ray") if wide_batch else None,
            "scene_bytes_detail": mem})

    # ---- first-timed-step replay (verdict r3 item 4a): the samples of the FIRST timed step again, untimed, same schedule (one wide batch), counters off;
    # the oracle checks >= 32 tiles of it whatever --steps is -- the whole-region gate below covers fewer tiles the more samples the timed region holds
    step0_hdr = None
    if rank == 0 and world == 1 and not args.no_parity:
        v.reset()
        v.render_tiles(tiles, warmup * spp_step, spp_step)
        step0_hdr = v.read_hdr()

    # ---- the reference's interactive regime (one Redraw() = +1 spp per call, AppViewer.cxx:1045-1047), reported beside `value`
    interactive = None
    if headline and rank == 0 and world == 1 and not args.no_interactive:
        interactive = interactive_figures(v, sc.camera)
    elif rank == 0 and world == 1 and not args.no_interactive:
        # the other legs: the lone frame after a restart only (crh_reset + crh_render(1) + crh_sync, median of 9) -- what a user of THIS scene waits for
        import statistics
        for _ in range(96):                                   # the library's own measurements for this scene come first (crh_get_frame_tuning: <= 30 frames; crh_get_tile_order: ~ 16 more)
            ft = v.frame_tuning(); v.tile_order()
            if (not ft["enabled"] or ft["feeders"]) and v.tile_order_calls["verdict"] != 0: break
            v.reset(); v.Redraw(); v.sync()
        ts = []
        for _ in range(10):
            v.reset(); v.sync()
            t1 = time.perf_counter(); v.Redraw(); v.sync()
            ts.append((time.perf_counter() - t1) * 1e3)
        interactive = {"first_frame_after_a_restart_ms": round(statistics.median(ts[1:]), 3), "first_frame_after_a_restart_ms_min": round(min(ts[1:]), 3),
                       "frame_feeders": v.frame_tuning()["feeders"], "tile_order": dict(v.tile_order_calls)}

    v.close()                                                 # the path state (up to 105 GB) and the scene go before the next leg / the CPU legs

    # ---- parity gates ("parity gate accompanying every number"; the reference's own gate is pixel-exact,
    # testing/CADRays_Testing.py:226-230): the oracle renders the SAME samples the timed steps rendered on sampled tiles; those pixels of
    # the accumulator the timed region left behind must be bit-identical (rel L2 <= 1e-4 is north_star's bar)
    parity = parity0 = None
    failed = False
    wide = (spp_step * len(tiles) * sc.params.tile_size ** 2) > (12 << 20)
    sched = ("big-batch (wide), counters off" if wide else "small-batch, counters off")
    if rank == 0 and (timed_hdr is not None or step0_hdr is not None):
        try:
            gate = ParityOracle(sc)
            if timed_hdr is not None:
                # at least 4 tiles whatever the region's length (20 steps x 512 samples: one tile costs the oracle ~7 s on 16 threads); the first-step gate
                # below keeps its >= 32 tiles
                parity = gate.check(timed_hdr, timed_first, timed_n, args.parity_seconds, 4)
                parity["schedule"] = sched + " -- the timed steps' own output"
            if step0_hdr is not None:
                parity0 = gate.check(step0_hdr, timed_first, spp_step, args.step0_seconds, 32)
                parity0["schedule"] = sched + " -- the FIRST timed step's samples replayed untimed after the timed region"
            gate.close()
        except Exception as e:                                  # a broken checker is its own state, not a failed gate: the measurement stands
            errors.append(f"{config} parity checker: {type(e).__name__}: {e}")
            if parity is None and timed_hdr is not None:
                parity = {"error": f"{type(e).__name__}: {e}"}
            if parity0 is None and step0_hdr is not None:
                parity0 = {"error": f"{type(e).__name__}: {e}"}
        for p in (parity, parity0):
            if p is not None and "error" not in p and not (p["pixels"] > 0 and p["rel_l2"] <= 1e-4):
                failed = True                                   # differing pixels -- or no pixel compared at all

    # ---- the gates of the interactive figures: the drag's last frame and a displayed frame, whole 1080p frames against the oracle
    if interactive and "_frames_for_gates" in interactive:
        fg = interactive.pop("_frames_for_gates")
        if not args.no_parity:
            try:
                import dataclasses as _dc
                import numpy as np
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import bench_redraw
                from oracle import pyoracle
                Or = pyoracle.oracle_class("parity"); Or.set_threads(usable_cpus())
                t1 = time.perf_counter()
                o = Or().load_scene(_dc.replace(sc, camera=bench_redraw.drag_camera(sc.camera, fg["drag_camera_index"])))
                o.render(1)
                drag_ok = bool(np.array_equal(o.read_hdr().view(np.uint32), fg["drag_hdr"].view(np.uint32)))
                o.set_camera(sc.camera); o.reset(); o.render(fg["displayed_frame"] + 1)
                shown_ok = bool(np.array_equal(o.read_ldr(), fg["displayed_ldr"]))
                o.close()
                interactive["gates"] = {"drag_last_frame_bit_exact": drag_ok, "displayed_frame_bit_exact": shown_ok,
                                        "what": f"whole {sc.params.width}x{sc.params.height} frames: the drag loop's frame {fg['drag_camera_index']} (HDR, 1 sample) and displayed frame "
                                                f"{fg['displayed_frame']} of the still camera (LDR through crh_read_ldr_begin / _end, {fg['displayed_frame'] + 1} samples) against the CPU oracle, "
                                                f"{time.perf_counter() - t1:.1f} s"}
                if not (drag_ok and shown_ok):
                    failed = True
            except Exception as e:
                errors.append(f"{config} interactive gates: {type(e).__name__}: {e}")
                interactive["gates"] = {"error": f"{type(e).__name__}: {e}"}
        rpf = interactive.get("rays_per_1spp_frame") or 0
        if rpf:
            if interactive.get("displayed_frames_per_s"):
                interactive["displayed_grays_per_s"] = round(interactive["displayed_frames_per_s"] * rpf / 1e9, 3)
            if interactive.get("first_frame_after_a_restart_ms"):
                interactive["first_frame_grays_per_s"] = round(rpf / (interactive["first_frame_after_a_restart_ms"] * 1e-3) / 1e9, 3)
            if interactive.get("drag_frames_per_s"):
                interactive["drag_grays_per_s"] = round(interactive["drag_frames_per_s"] * rpf / 1e9, 3)

    # ---- CPU baseline: the oracle (a port, not the reference: OCCT has no CPU path tracer) on this box's cores
    cpu = None
    if headline and rank == 0 and not args.no_cpu and world == 1:          # reported on rank 0 at N = 1 only
        try:
            cpu = cpu_baseline(sc, args.cpu_seconds)
        except Exception as e:
            cpu = {"value": None, "error": f"{type(e).__name__}: {e}"}

    out = None
    if rank == 0:
        mrays = (rays_n + rays_a) / dt / 1e6
        cfgd = {"workload": f"{config}: {len(sc.tri)} {'triangles of the Cornell box (CornellBox.tcl without the spheres)' if scene_cfg == 'C1' else ('triangles of a tessellated CAD-like assembly (scenes.gen_cad_like: shared vertices, long thin triangles, coincident faces)' if scene_cfg == 'CAD1M' else 'random triangles')}, {len(sc.materials)} BSDF(s), "
                            f"{'HDR sky env' if sc.env is not None else 'constant env'}, {len(sc.lights)} light(s), "
                            f"{sc.params.width}x{sc.params.height}, depth {sc.params.max_depth}",
                "spp_per_step_per_rank": spp_step, "spp_per_step_whole_frame": spp_step if scaling == "strong" or world == 1 else spp,
                "tiles_per_rank": int(len(tiles)), "parallelism": f"tiles x{n_gpus} + {assemble_info['mode'] if assemble_info else 'RCCL reduce'}" if n_gpus > 1 else "single GPU",
                "rccl_ranks": int(rccl_ranks), "backend": backend,
                "msamples_per_s": round(samples / dt / 1e6, 3), "rays_nearest": int(rays_n), "rays_any": int(rays_a),
                "build_upload_s": round(build_s, 2), "scene_generation_s": round(gen_s, 2), "host_cores": os.cpu_count(), "host_usable_cpus": usable_cpus(), "interactive": interactive}
        if n_gpus > 1:
            rr = [p["rays"] for p in per_rank]
            cfgd.update({"per_rank": per_rank, "reduce_ms_per_step": max(p["ms_reduce"] for p in per_rank),
                         "render_ms_per_step_max": max(p["ms_render"] for p in per_rank),
                         "shard_rays_max_over_mean": round(max(rr) / max(sum(rr) / len(rr), 1.0), 4),
                         "assemble": assemble_info, "scene_hand_over": share,
                         "rccl_version": ".".join(str(x) for x in torch.cuda.nccl.version()) if backend == "nccl" else None,
                         "per_rank_note": "ms_render = crh_render_tiles + sync per timed step on that rank, ms_reduce = the exchange step as that rank saw it (it includes waiting for slower ranks)"})
        out = {
            "metric": "Mrays/s", "value": round(mrays, 2), "unit": "Mrays/s", "n_gpus": n_gpus, "steps": steps, "warmup": warmup,
            "ms_per_step": round(dt / steps * 1e3, 3), "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "f32", "data": "synthetic", "config": cfgd,
            "roofline": roof, "cpu_baseline": cpu, "parity": parity, "parity_step0": parity0,
        }
    return out, failed, errors


def interactive_figures(v, cam0):
    interactive = {}
    # the first 30 frame-kernel frames after a build are the library's measurement of its feeder count (crh_get_frame_tuning): a viewer passes them in the first
    # tenth of a second; the figures below are those of the settled library
    v.set_lookahead(1)
    for _ in range(96):                                       # (the feeder count, then whether the sorted tile list pays on this scene: crh_get_tile_order)
        ft = v.frame_tuning(); v.tile_order()
        if (not ft["enabled"] or ft["feeders"]) and v.tile_order_calls["verdict"] != 0: break
        v.reset(); v.Redraw(); v.sync()
    interactive["frame_tuning"] = v.frame_tuning(); interactive["tile_order"] = dict(v.tile_order_calls)
    for k in (1, 16, 64):
        v.set_lookahead(k); v.reset()
        for _ in range(max(8, 2 * k)):              # the frame pipeline (up to eight in flight) is full before the clock starts
            v.Redraw()
        v.sync()
        n_fr = max(64, 4 * k)
        t1 = time.perf_counter()
        for _ in range(n_fr):
            v.Redraw()
        v.sync()
        interactive[f"redraw_per_s_lookahead_{k}"] = round(n_fr / (time.perf_counter() - t1), 1)
    v.set_lookahead(1)
    # crh_set_lookahead_auto(16): FROM a restart -- one sample, then batches of 4, 16, 16, ... -- 64 Redraw()s, three sessions
    v.set_lookahead_auto(16); v.reset(); v.sync()
    t1 = time.perf_counter()
    for _ in range(3):
        v.reset()
        for _ in range(64):
            v.Redraw()
    v.sync()
    interactive["redraw_per_s_lookahead_auto_16_first_64_frames_after_a_restart"] = round(3 * 64 / (time.perf_counter() - t1), 1)
    v.set_lookahead_auto(0); v.reset()
    # ---- the application's own call pattern (AppViewer.cxx:979-984, 1045-1047, 1099): a lone frame after a restart (median of 9), the restart-every-frame
    # drag, every frame displayed (asynchronous LDR read-back two frames behind) -- tools/bench_redraw.py holds the loops
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_redraw
    interactive.update(bench_redraw.measure(v, cam0, frames=96, trials=9))
    interactive.pop("free_running_redraw_per_s", None)          # = redraw_per_s_lookahead_1 above
    # ---- what the gates of the interactive figures compare (round-5 verdict, item 8): the LAST frame of the drag loop above (camera of frame 95, one
    # sample) and a DISPLAYED frame of the still camera (the fourth: 4 samples, through the asynchronous LDR read-back) -- the oracle renders both after
    # the view is closed (run_leg); rays of a 1-sample frame turn the frame rates into Grays/s
    v.set_camera(bench_redraw.drag_camera(cam0, 95)); v.reset(); v.Redraw()
    frames_for_gates = {"drag_hdr": v.read_hdr(), "drag_camera_index": 95}
    v.set_camera(cam0); v.reset(); v.sync()
    shown = []
    for i in range(4):
        v.Redraw()
        if i >= 2: shown.append(v.read_ldr_end())
        v.read_ldr_begin()
    shown.append(v.read_ldr_end()); shown.append(v.read_ldr_end())
    frames_for_gates["displayed_ldr"] = shown[3]; frames_for_gates["displayed_frame"] = 3
    v.reset(); v.Redraw(); s1 = v.stats()
    interactive["rays_per_1spp_frame"] = int(s1["rays_nearest"] + s1["rays_any"])
    interactive["_frames_for_gates"] = frames_for_gates
    v.reset()
    import cadrays_amd
    frames, queues = cadrays_amd.pipeline_capacity()
    interactive["frames_in_flight"] = frames
    interactive["hw_queues"] = queues
    interactive["note"] = ("one crh_render(1) per call over the whole frame, no read-back; NOT part of `value`; bench.py (the host) exported GPU_MAX_HW_QUEUES before "
                           "the first HIP call -- the library itself never writes the environment (crh_query_pipeline_capacity)")
    return interactive


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def usable_cpus():
    from cadrays_amd.hostinfo import usable_cpus as f       # pure Python: the launcher parent still never imports torch
    return f()


def cpu_baseline(sc, seconds):
    """The CPU oracle on a bounded tile sample of the same workload: the parity build (what the tests compare against) and,
    when built, the fast build (-O3 -march=native, contraction allowed, counters compiled out; oracle/Makefile)."""
    import numpy as np
    from oracle import pyoracle
    ncores = usable_cpus()
    out = None
    for kind in ("fast", "parity"):
        try:
            Or = pyoracle.oracle_class(kind)
        except (OSError, AttributeError, RuntimeError, subprocess.SubprocessError):      # e.g. a failing -march=native build on this host
            continue
        Or.set_threads(ncores)
        o = Or().load_scene(sc)
        nt = o.n_tiles()
        budget = seconds if kind == "fast" or out is None else max(4.0, seconds / 3)
        sample = np.unique(np.linspace(0, nt - 1, 32).astype(np.uint32))
        o.render_tiles(sample, 0, 1)                      # calibration (also warms the threads)
        s0 = o.stats()
        rate = (s0["rays_nearest"] + s0["rays_any"]) / max(s0["seconds"], 1e-9)
        per_tile = (s0["rays_nearest"] + s0["rays_any"]) / len(sample)
        want = int(min(nt, max(32, rate * budget / max(per_tile, 1))))
        sample = np.unique(np.linspace(0, nt - 1, want).astype(np.uint32))
        # first pass at 1 spp re-measures the rate with all threads busy; then size spp for ~budget seconds
        o.reset(); o.render_tiles(sample, 0, 1)
        s1 = o.stats()
        rate = (s1["rays_nearest"] + s1["rays_any"]) / max(s1["seconds"], 1e-9)
        spp_cpu = int(max(1, min(64, rate * budget / max(s1["rays_nearest"] + s1["rays_any"], 1))))
        o.reset(); o.render_tiles(sample, 0, spp_cpu)
        s1 = o.stats()
        leg = {"value": round((s1["rays_nearest"] + s1["rays_any"]) / s1["seconds"] / 1e6, 3), "unit": "Mrays/s",
               "sample": f"{len(sample)} of {nt} 32x32 tiles of the same workload, {spp_cpu} spp, {s1['seconds']:.1f} s, OpenMP oracle ({kind} build)"}
        if out is None:
            out = {"value": leg["value"], "unit": "Mrays/s", "cores": ncores, "kind": "port", "sample": leg["sample"],
                   "cpu_model": cpu_model(), "build": kind, "hardware_threads": os.cpu_count(),
                   "cores_note": "threads used = CPUs this container may run on at once (affinity mask and cgroup CPU quota)"}
        out[f"{kind}_build"] = leg
        o.close()
    return out


class ParityOracle:
    """The CPU oracle (parity build: -O2, no contraction) loaded once per leg.  check() renders samples [first_sample, first_sample + n_samples) of a
    bounded, evenly spread tile sample of the same workload -- the samples the timed steps rendered -- and the GPU frame must hold the same bits there."""

    def __init__(self, sc):
        from oracle import pyoracle
        Or = pyoracle.oracle_class("parity")
        Or.set_threads(usable_cpus())
        t0 = time.perf_counter()
        self.o = Or().load_scene(sc)
        self.load_s = time.perf_counter() - t0
        self.rate = None

    def close(self):
        self.o.close()

    def check(self, gpu_hdr, first_sample, n_samples, seconds, min_tiles):
        import numpy as np
        o = self.o
        nt = o.n_tiles()
        if self.rate is None:
            probe = np.unique(np.linspace(0, nt - 1, 16).astype(np.uint32))
            o.reset(); o.render_tiles(probe, first_sample, 1)          # calibration: rays per tile-sample and the rate on this host
            s0 = o.stats()
            self.per_tile_sample = (s0["rays_nearest"] + s0["rays_any"]) / len(probe)
            self.rate = (s0["rays_nearest"] + s0["rays_any"]) / max(s0["seconds"], 1e-9)
        want = int(max(min_tiles, min(256, nt, self.rate * seconds / max(self.per_tile_sample * n_samples, 1))))
        k = min(want, nt)
        sample = np.unique(((np.arange(k) + 0.5) * nt / k).astype(np.uint32))          # the middle of k equal runs of the tile list (not its ends: the corners of a frame are mostly sky)
        o.reset()
        t0 = time.perf_counter()
        o.render_tiles(sample, first_sample, n_samples)
        dt = time.perf_counter() - t0
        ref = o.read_accum()
        mask = ref[..., 3] == n_samples
        g = np.ascontiguousarray(gpu_hdr[mask], np.float32)
        r = np.ascontiguousarray(ref[..., :3][mask], np.float32)
        rel = float(np.linalg.norm(g.astype(np.float64) - r) / max(np.linalg.norm(r.astype(np.float64)), 1e-300))
        diff = int((g.view(np.uint32) != r.view(np.uint32)).sum())
        return {"rel_l2": rel, "bit_exact": diff == 0 and int(mask.sum()) > 0, "words_differing": diff, "tiles": int(len(sample)), "pixels": int(mask.sum()),
                "spp": int(n_samples), "first_sample": int(first_sample), "tolerance_rel_l2": 1e-4,
                "oracle": f"CPU oracle, parity build, {usable_cpus()} threads, {dt:.1f} s (scene hand-over {self.load_s:.1f} s)"}


if __name__ == "__main__":
    main()
