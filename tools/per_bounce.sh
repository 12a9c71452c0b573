cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o t -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu --other-configs none $BENCH_ARGS > /tmp/kt.log 2>&1
tail -5 /tmp/kt.log; find /tmp/kt -name "*.csv" | head; f=$(ls /tmp/kt/*/*kernel_trace.csv /tmp/kt/*kernel_trace.csv 2>/dev/null | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
out = []
for r in rows:
    n = r['Kernel_Name']
    import re; m = re.search(r'k_[a-z_]+', n); short = m.group(0) if m else n[:40]
    out.append((short, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6, int(r['Start_Timestamp']), int(r['End_Timestamp'])))
# print the launches of the timed step: find the second k_raygen
idx = [i for i, o_ in enumerate(out) if 'k_raygen' in o_[0]]
print('raygen launches at', idx)
s = idx[1]; e = idx[2] if len(idx) > 2 else len(out)
prev_end = None
for n, ms, st, en in out[s:e]:
    print('%-42s %8.3f ms   gap before %7.1f us' % (n, ms, (st - prev_end) / 1e3 if prev_end else 0.0))
    prev_end = en
PY
