#!/usr/bin/env python3
"""Summarise rocprofv3 CSV outputs (kernel stats + PMC per kernel) into a short text table."""
import csv, glob, os, sys, collections
out = sys.argv[1]
def short(n):
    for tok in ("crh::(anonymous namespace)::", "void ", "crh::"):
        n = n.replace(tok, "")
    return n.split("(")[0][:40]
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    print("== kernel stats:", os.path.relpath(f, out))
    for r in csv.DictReader(open(f)):
        print(f"{short(r['Name']):60s} calls={r['Calls']:>6s} total_ns={r['TotalDurationNs']:>14s} avg_ns={float(r['AverageNs']):>12.0f} pct={r['Percentage']}")
for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    if not os.path.isdir(d): continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"]); agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
        print("== pmc:", os.path.basename(d))
        for k in agg:
            print(f"{k:60s} " + " ".join(f"{c}={v:.4g} (n={cnt[(k,c)]}, per-launch {v/cnt[(k,c)]:.4g})" for c, v in agg[k].items()))
