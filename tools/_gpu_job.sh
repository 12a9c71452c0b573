mkdir -p gpurun_out/r2i
python tools/bench_interactive.py 2>/dev/null | tail -1 | tee gpurun_out/r2i/interactive.txt
CRH_LANES=1 python tools/bench_interactive.py 2>/dev/null | tail -1 | tee -a gpurun_out/r2i/interactive.txt
python tools/bench_interactive.py --config C2 2>/dev/null | tail -1 | tee -a gpurun_out/r2i/interactive.txt
python tools/bench_transforms.py 10 2>/dev/null | tail -1 | tee gpurun_out/r2i/bench_transforms.txt
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r2i/pytest_gpu.log 2>&1; grep -E "^FAILED|passed|failed" gpurun_out/r2i/pytest_gpu.log | cut -c1-200
