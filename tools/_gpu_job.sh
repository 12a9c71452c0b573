mkdir -p gpurun_out/r2c
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r2c/pytest_gpu.log 2>&1; grep -E "^FAILED|passed|failed" gpurun_out/r2c/pytest_gpu.log | cut -c1-200
timeout 1500 python -m pytest tests -m gpu -q -p no:randomly > gpurun_out/r2c/pytest_gpu2.log 2>&1; grep -E "^FAILED|passed|failed" gpurun_out/r2c/pytest_gpu2.log | cut -c1-200
python tools/bench_transforms.py 10 30 2>&1 | tee gpurun_out/r2c/bench_transforms.txt
python bench.py --no-cpu 2>/dev/null | tail -1 > gpurun_out/r2c/bench_C3.json; cut -c1-200 gpurun_out/r2c/bench_C3.json; grep -o '"interactive": {[^}]*}' gpurun_out/r2c/bench_C3.json
