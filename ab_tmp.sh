python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_two_level.py -x -q -m gpu > gpurun_out/t.log 2>&1; grep -E "passed|failed" gpurun_out/t.log
for i in 1 2; do python bench.py --steps 3 --warmup 1 2>&1 | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); r = j['roofline']; print(j['value'], j['ms_per_step'], r['avg_launch_ms'], r['achieved'])
"; done
python bench.py --config C2 --steps 3 --warmup 1 2>&1 | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); print('C2', j['value'], j['ms_per_step'])
"
