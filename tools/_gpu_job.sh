echo "== parity (pipeline default on)"
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_golden.py tests/test_two_level.py tests/test_adaptive.py -m gpu -x -q 2>&1 | grep -E "passed|failed|FAILED" | head -5
for pl in 0 1; do echo "== CRH_PIPELINE=$pl"; CRH_PIPELINE=$pl timeout 300 python tools/bench_interactive.py 2>/dev/null | tail -1 | cut -c60-330; done
for div in 1024 1365 4096; do echo "== PIPELINE=1 PIPE_DIV=$div"; CRH_PIPE_DIV=$div timeout 300 python tools/bench_interactive.py --frames 48 2>/dev/null | tail -1 | cut -c60-200; done
echo "== C2"; timeout 300 python tools/bench_interactive.py --config C2 2>/dev/null | tail -1 | cut -c60-330
