python tools/bench_transforms.py 10 2>/dev/null | tail -1
python tools/bench_interactive.py 2>/dev/null | tail -1 | cut -c1-260
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q 2>&1 | tail -1
