#!/usr/bin/env python3
"""Per-kernel per-launch averages of a profiles/frame_profile.sh folder:  python3 profiles/frame_fold.py gpurun_out/frame_<tag>"""
import collections, csv, glob, os, sys
src = sys.argv[1]
def short(n):
    for tok in ("crh::(anonymous namespace)::", "void ", "crh::"): n = n.replace(tok, "")
    return n.split("(")[0][:44]
print(open(os.path.join(src, "trace.log")).read().strip().splitlines()[-1])
for f in glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if float(r["Percentage"]) >= 0.5: print(f"{short(r['Name']):46s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs']) / 1e3:>9.1f} pct={r['Percentage']}")
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(os.path.join(src, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"]); tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k, d in tot.items():
    if "k_" not in k: continue
    a = {c: v / cnt[(k, c)] for c, v in d.items()}
    line = f"{k:46s} " + " ".join(f"{c.replace('SQ_', '')}={v:.3g}" for c, v in sorted(a.items()))
    print(line)
    if "SQ_INSTS_VALU" in a and a.get("SQ_ACTIVE_INST_VALU") and a.get("SQ_BUSY_CYCLES"):
        # units (profiles/r3/counter_units.md): INSTS_VALU wave-level instructions; THREAD_CYCLES / ACTIVE_INST = active lanes per instruction; BUSY_CYCLES one
        # count per shader engine (32): / 32 = shader cycles of the launch; 1024 SIMDs issue one wave64 VALU instruction per 4 cycles
        cyc = a["SQ_BUSY_CYCLES"] / 32.0
        print(f"{'':46s}   lanes/VALU instr {a['SQ_THREAD_CYCLES_VALU'] / a['SQ_ACTIVE_INST_VALU']:.1f}  VALU issue {a['SQ_INSTS_VALU'] * 4 / (1024 * cyc):.2f} of the SIMD cycles  "
              f"launch {cyc / 2.4e3:.0f} us at 2.4 GHz  waves {a.get('SQ_WAVES', 0):.0f}  wave-cycles/launch-cycles {a.get('SQ_WAVE_CYCLES', 0) * 4 / max(a.get('SQ_WAVES', 1), 1) / cyc:.2f}")
