mkdir -p gpurun_out/r2j
python tools/bench_interactive.py 2>/dev/null | tail -1 | tee gpurun_out/r2j/interactive.txt
timeout 1200 python tools/big_fuzz.py 2000 2400 2>&1 | tail -1 | tee gpurun_out/r2j/big_fuzz.txt
timeout 900 python bench.py --config C4 --no-cpu --no-interactive 2>/dev/null | tail -1 | tee gpurun_out/r2j/bench_C4.json | cut -c1-400
