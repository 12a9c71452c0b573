cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ad -o t -- python3 $GRAFT_REPO_ROOT/tools/_adaptive_probe.py > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/ad/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f))); rows.sort(key=lambda r: int(r['Start_Timestamp']))
def short(n):
    for k in ('k_raygen','k_trace_nearest','k_trace_any','k_shade','k_accumulate','k_tile_error','k_adaptive_pick','fillBuffer','copyBuffer'):
        if k in n: return k
    return n[:30]
frames=[]; cur=[]
for r in rows:
    if 'k_tile_error' in r['Kernel_Name'] and cur: frames.append(cur); cur=[]
    cur.append(r)
fr=[f for f in frames if len(f)==len(frames[20])][10:30]
agg=collections.defaultdict(list); spans=[]; gaps=[]
for f_ in fr:
    spans.append((int(f_[-1]['End_Timestamp'])-int(f_[0]['Start_Timestamp']))/1e3)
    for i,r in enumerate(f_):
        agg[(i,short(r['Kernel_Name']))].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
        if i: gaps.append((int(r['Start_Timestamp'])-int(f_[i-1]['End_Timestamp']))/1e3)
print("iteration span us %.0f, launches %d, gap total %.0f"%(sum(spans)/len(spans), len(fr[0]), sum(gaps)/len(fr)))
print(" ".join("%s:%.0f"%(k[1].replace('k_',''),sum(v)/len(v)) for k,v in sorted(agg.items())))
starts=[int(f_[0]['Start_Timestamp']) for f_ in fr]
print("period us", (starts[-1]-starts[0])/1e3/(len(starts)-1))
PY
