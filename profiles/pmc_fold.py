#!/usr/bin/env python3
"""Fold the rocprofv3 CSVs of profiles/pmc_collect.sh into profiles/pmc_traffic.json.

  python3 profiles/pmc_fold.py gpurun_out/pmc_<tag> C3 C2 C5

Per config: per-launch averages of FETCH_SIZE, WRITE_SIZE, TCC_HIT_sum, TCC_MISS_sum of the traversal launches of the timed path -- k_trace_nearest<false, *>
and, for the camera rays of wide batches, k_trace_packets with its fall-back pass (the counting pass of bench.py runs <true, *>) -- and their average duration from the kernel trace.
  hbm_bytes_per_launch = FETCH_SIZE * 1024 * 2 + WRITE_SIZE * 1024
(the x2 on the read side is the gfx950 correction of /opt/skills/guides/MI355X_MICROARCH.md: FETCH_SIZE tallies 128-B requests
at 64 B; cross-check: TCC_MISS_sum * 128 B).  Memory-side counter: Infinity-Cache hits are included, so this is an UPPER bound
on HBM bytes.  Every entry carries the hash of the kernel sources it was measured on (bench.kernel_source_hash)."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (hash + file location only; imports nothing GPU-related)


def timed_kernel(name):
    """a traversal launch of the TIMED path: k_trace_nearest<COUNT = false, ...> of bounces 1 .., and k_trace_packets, which walks the camera rays (bounce 0) of
    wide batches since round 4.  Its fall-back pass (k_trace_nearest<.., FB = true>: the few rays that met a tie) belongs to the bounce-0 launch: its counters
    are added, it is not counted as a launch of its own (fallback_kernel)."""
    n = name.replace(" ", "").replace("(bool)0", "false").replace("(bool)1", "true").replace("<0", "<false")
    return "k_trace_nearest<false" in n or "k_trace_packets" in n


def fallback_kernel(name):
    n = name.replace(" ", "").replace("(bool)0", "false").replace("(bool)1", "true")
    return "k_trace_nearest<false" in n and n.split("k_trace_nearest<")[1].split(">")[0].split(",")[-1] == "true" and n.split("k_trace_nearest<")[1].split(">")[0].count(",") == 4


def counters(d):
    agg, cnt = collections.defaultdict(float), collections.Counter()
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if timed_kernel(r["Kernel_Name"]):
                agg[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 0 if fallback_kernel(r["Kernel_Name"]) else 1
    return {k: agg[k] / cnt[k] for k in agg}, dict(cnt)


def main():
    out, cfgs = sys.argv[1], sys.argv[2:]
    path = bench.PMC_FILE
    data = {"configs": {}}
    if os.path.exists(path):
        old = json.load(open(path))
        if "configs" in old:
            data = old
    data["kernel"] = "k_trace_nearest"
    data["hash_of"] = list(bench.pmc_hash_files())
    for cfg in cfgs:
        vals = {}
        for d in sorted(glob.glob(os.path.join(out, cfg, "pmc_*"))):
            v, n = counters(d)
            vals.update(v)
        if "FETCH_SIZE" not in vals or "WRITE_SIZE" not in vals:
            print(f"{cfg}: counters missing ({sorted(vals)})"); continue
        avg_ms = None
        for f in glob.glob(os.path.join(out, cfg, "trace", "**", "*kernel_stats.csv"), recursive=True):
            tot_ns, calls = 0.0, 0
            for r in csv.DictReader(open(f)):
                if timed_kernel(r["Name"]):
                    tot_ns += float(r["TotalDurationNs"]); calls += 0 if fallback_kernel(r["Name"]) else int(r["Calls"])
            if calls:
                avg_ms = tot_ns / calls * 1e-6
        bj = {}
        try:
            bj = json.loads(open(os.path.join(out, cfg + ".bench.json")).read())
        except (OSError, ValueError):
            pass
        ent = {"source_hash": bench.kernel_source_hash(),
               "spp_per_step": (bj.get("config") or {}).get("spp_per_step_per_rank"),
               "workload": (bj.get("config") or {}).get("workload"),
               "fetch_size_kb_per_launch": vals["FETCH_SIZE"], "write_size_kb_per_launch": vals["WRITE_SIZE"],
               "tcc_hit_per_launch": vals.get("TCC_HIT_sum"), "tcc_miss_per_launch": vals.get("TCC_MISS_sum"),
               "hbm_bytes_per_launch": vals["FETCH_SIZE"] * 1024 * 2 + vals["WRITE_SIZE"] * 1024,
               "tcc_miss_x128_bytes_per_launch": vals["TCC_MISS_sum"] * 128 if "TCC_MISS_sum" in vals else None,
               "avg_launch_ms": avg_ms, "from": os.path.basename(os.path.normpath(out)),
               # everything else that was collected, per launch of the timed instantiation (SQ_* activity counters are in quad-cycles summed over
               # the SIMDs; see bench.roofline_ceilings for what is derived from them)
               "counters_per_launch": {k: vals[k] for k in sorted(vals) if k not in ("FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum")},
               "how": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / TCC_HIT_sum TCC_MISS_sum / TCC_REQ_sum TCC_READ_sum / SQ groups in separate passes of `bench.py --config %s --steps 2 --warmup 1 --no-cpu --no-interactive --no-parity --other-configs none` (profiles/pmc_collect.sh); bytes = FETCH_SIZE*1024*2 + WRITE_SIZE*1024 (gfx950 x2 read-side correction); memory-side counter, Infinity-Cache hits included" % cfg}
        data["configs"][cfg] = ent
        print(cfg, json.dumps(ent))
    json.dump(data, open(path, "w"), indent=1)
    # keep a copy beside the raw CSVs so that the gpurun_out merge brings it back
    json.dump(data, open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
