#!/usr/bin/env python3
"""Per-kernel timeline of one 1-spp Redraw() from a rocprofv3 --kernel-trace CSV of tools/bench_interactive.py:
   python3 tools/frame_timeline.py <t_kernel_trace.csv>"""
import collections, csv, statistics, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
def short(n):
    for k in ('k_raygen', 'k_trace_nearest', 'k_trace_any', 'k_shade', 'k_accumulate', 'k_tonemap', 'k_hdr', 'fillBuffer', 'copyBuffer', 'k_tile_error', 'k_adaptive_pick'):
        if k in n: return k
    return n[:30]
frames, cur = [], []
for r in rows:
    if 'k_raygen' in r['Kernel_Name'] and cur: frames.append(cur); cur = []
    cur.append(r)
lens = collections.Counter(len(f) for f in frames)
L = lens.most_common(1)[0][0]
fr_ = [f for f in frames if len(f) == L][10:30]
agg = collections.defaultdict(list); spans = []
for fr in fr_:
    spans.append((int(fr[-1]['End_Timestamp']) - int(fr[0]['Start_Timestamp'])) / 1e3)
    for i, r in enumerate(fr):
        agg[(i, short(r['Kernel_Name']))].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
print("launches per frame", L, " frame span us: mean %.1f" % (sum(spans) / len(spans)))
print(" ".join("%s:%.0f" % (k[1].replace('k_', ''), sum(v) / len(v)) for k, v in sorted(agg.items())))
