mkdir -p gpurun_out/r2g
python bench.py 2> gpurun_out/r2g/bench_C3.err | tail -1 > gpurun_out/r2g/bench_C3.json; cat gpurun_out/r2g/bench_C3.json
bash profiles/pmc_collect.sh r2g C3 C2 C5 > gpurun_out/r2g/pmc.log 2>&1; tail -3 gpurun_out/r2g/pmc.log | cut -c1-300
