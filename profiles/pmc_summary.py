#!/usr/bin/env python3
"""Text summary of one profiles/pmc_collect.sh output folder:  python3 profiles/pmc_summary.py gpurun_out/pmc_<tag> <dest dir>
Per config: rocprofv3 --kernel-trace --stats table (calls, total, average, share) and the per-launch averages of every counter
of every kernel; copies the kernel_stats CSVs and the bench lines next to it."""
import collections, csv, glob, os, shutil, sys

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(dst, exist_ok=True)


def short(n):
    for tok in ("crh::(anonymous namespace)::", "void ", "crh::"):
        n = n.replace(tok, "")
    return n.split("(")[0][:58]


for cfg in sorted(d for d in os.listdir(src) if os.path.isdir(os.path.join(src, d))):
    out = open(os.path.join(dst, f"rocprof_summary_{cfg}.txt"), "w")
    for f in glob.glob(os.path.join(src, cfg, "trace", "**", "*kernel_stats.csv"), recursive=True):
        shutil.copyfile(f, os.path.join(dst, f"kernel_stats_{cfg}.csv"))
        out.write(f"== {cfg}: rocprofv3 --kernel-trace --stats of `bench.py --config {cfg} --steps 2 --warmup 1 --no-cpu --no-interactive`\n")
        for r in csv.DictReader(open(f)):
            out.write(f"{short(r['Name']):60s} calls={r['Calls']:>5s} total_ns={r['TotalDurationNs']:>13s} avg_ns={float(r['AverageNs']):>12.0f} pct={r['Percentage']}\n")
    for d in sorted(glob.glob(os.path.join(src, cfg, "pmc_*"))):
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"]); agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
            out.write(f"== {cfg}: --pmc {os.path.basename(d)[4:].replace('_', ' ')} (separate pass)\n")
            for k in agg:
                out.write(f"{k:60s} " + " ".join(f"{c}={v:.4g} (n={cnt[(k, c)]}, per launch {v / cnt[(k, c)]:.4g})" for c, v in agg[k].items()) + "\n")
    b = os.path.join(src, cfg + ".bench.json")
    if os.path.exists(b):
        shutil.copyfile(b, os.path.join(dst, f"bench_{cfg}_under_rocprof.json"))
    out.close()
    print(open(os.path.join(dst, f"rocprof_summary_{cfg}.txt")).read())
