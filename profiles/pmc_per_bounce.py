#!/usr/bin/env python3
"""Per-launch view of the counter passes profiles/pmc_collect.sh already takes (round-3 verdict item 2c: which bounce owns the idle lanes; item 6: what the
streaming kernels move).  rocprofv3 --pmc writes one row per dispatch; pmc_fold.py averages the timed traversal kernel over all of them -- this prints them
one by one, in launch order, for ONE timed step of the bench command:

  traversal   per bounce: launch time, VALU instructions, lane utilisation = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU), VALU issue share =
              SQ_ACTIVE_INST_VALU x (SQ_WAVES / 1024) / SQ_WAVE_CYCLES (units: profiles/r3/counter_units.md), memory-side GB (FETCH_SIZE x 1024 x 2 + WRITE_SIZE x 1024)
  streaming   k_raygen / k_shade per bounce / k_accumulate: launch time, read + written GB, TB/s against the 8 TB/s peak

  python3 profiles/pmc_per_bounce.py gpurun_out/pmc_<tag> C3 [C5 ...] > profiles/r4/per_bounce_counters.txt
Launch times are the profiled ones (counter collection slows a launch by 5-10 %); the ratios are what this is for."""
import collections
import csv
import glob
import os
import sys


def rows_of(d, want):
    out = collections.OrderedDict()
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            k = next((w for w in want if w in n), None)
            if k is None:
                continue
            e = out.setdefault(int(r["Dispatch_Id"]), {"kernel": k, "name": n, "ms": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6})
            e[r["Counter_Name"]] = float(r["Counter_Value"])
    return list(out.values())


def one_step(rows, first_kernel):
    """the launches of the SECOND batch (k_raygen .. the next k_raygen) that contains a timed traversal launch (the first belongs to the warm-up step; the counting pass comes last)"""
    idx = [i for i, r in enumerate(rows) if r["kernel"] == first_kernel] + [len(rows)]
    steps = [rows[idx[k]:idx[k + 1]] for k in range(len(idx) - 1)]
    good = [st for st in steps if any(r["kernel"] in ("k_trace_nearest", "k_trace_packets") for r in st)]
    return good[1] if len(good) > 1 else (good[0] if good else rows)


def timed(name):
    n = name.replace(" ", "").replace("(bool)0", "false").replace("(bool)1", "true")
    return "<false" in n or "<0" in n


def main():
    root, cfgs = sys.argv[1], sys.argv[2:]
    want = ("k_raygen", "k_trace_packets", "k_trace_nearest", "k_shade", "k_trace_any", "k_accumulate")
    for cfg in cfgs:
        passes = {os.path.basename(d)[4:]: rows_of(d, want) for d in sorted(glob.glob(os.path.join(root, cfg, "pmc_*")))}
        sq = next((v for k, v in passes.items() if k.startswith("SQ_WAVES")), None)
        fetch, write = passes.get("FETCH_SIZE"), passes.get("WRITE_SIZE")
        print(f"== {cfg}: one batch (k_raygen .. k_accumulate; a step of C3 is two of them, of C5 four, of C2 / C1 one) of `bench.py --config {cfg} --steps 2 --warmup 1` (counting-pass launches excluded), launch order")
        if sq:
            step = [r for r in one_step([r for r in sq if r["kernel"] != "k_trace_nearest" or timed(r["name"])], "k_raygen")]
            f_step = one_step([r for r in (fetch or []) if r["kernel"] != "k_trace_nearest" or timed(r["name"])], "k_raygen")
            w_step = one_step([r for r in (write or []) if r["kernel"] != "k_trace_nearest" or timed(r["name"])], "k_raygen")
            print(f"{'kernel':18s} {'bounce':>6s} {'ms':>8s} {'VALU insts':>12s} {'lane util':>10s} {'VALU issue':>11s} {'read GB':>9s} {'write GB':>9s} {'TB/s':>7s} {'of 8 TB/s':>9s}")
            bounce = collections.Counter()
            for i, r in enumerate(step):
                grp = "trace" if r["kernel"] in ("k_trace_packets", "k_trace_nearest") else r["kernel"]
                if r["kernel"] == "k_trace_nearest" and i and step[i - 1]["kernel"] == "k_trace_packets":
                    b = bounce[grp] - 1                 # the fall-back pass of the packet launch: same bounce
                else:
                    b = bounce[grp]; bounce[grp] += 1
                lane = r["SQ_THREAD_CYCLES_VALU"] / (64.0 * r["SQ_ACTIVE_INST_VALU"]) if r.get("SQ_ACTIVE_INST_VALU") else float("nan")
                issue = r["SQ_ACTIVE_INST_VALU"] * (r["SQ_WAVES"] / 1024.0) / r["SQ_WAVE_CYCLES"] if r.get("SQ_WAVE_CYCLES") else float("nan")
                rd = f_step[i]["FETCH_SIZE"] * 1024 * 2 / 1e9 if i < len(f_step) and f_step[i]["kernel"] == r["kernel"] and "FETCH_SIZE" in f_step[i] else float("nan")
                wr = w_step[i]["WRITE_SIZE"] * 1024 / 1e9 if i < len(w_step) and w_step[i]["kernel"] == r["kernel"] and "WRITE_SIZE" in w_step[i] else float("nan")
                ms = f_step[i]["ms"] if i < len(f_step) and f_step[i]["kernel"] == r["kernel"] else r["ms"]
                tbs = (rd + wr) / ms if ms > 0 else float("nan")
                print(f"{r['kernel']:18s} {b:6d} {r['ms']:8.3f} {r.get('SQ_INSTS_VALU', float('nan')):12.4g} {lane:10.3f} {issue:11.3f} {rd:9.2f} {wr:9.2f} {tbs:7.2f} {tbs / 8.0:9.2f}")
            tot = collections.defaultdict(float)
            for r in step:
                tot[r["kernel"]] += r["ms"]
            print("batch total (profiled ms): " + ", ".join(f"{k} {v:.1f}" for k, v in tot.items()))
        print()


if __name__ == "__main__":
    main()
