mkdir -p gpurun_out/r2m
python bench.py 2>/dev/null | tail -1 > gpurun_out/r2m/bench_C3.json; cut -c1-130 gpurun_out/r2m/bench_C3.json
for cfg in C2 C5; do python bench.py --config $cfg --no-cpu 2>/dev/null | tail -1 > gpurun_out/r2m/bench_$cfg.json; cut -c1-130 gpurun_out/r2m/bench_$cfg.json; done
