#!/bin/bash
# One GPU-box session: GPU test suite, the bench lines of every config, rocprofv3 kernel stats + counter passes.
#   gpurun --timeout 3000 -- 'bash tools/gpu_round.sh <tag> [skip-tests]'
TAG=${1:-x}
OUT=gpurun_out/$TAG
mkdir -p $OUT
if [ "$2" != "skip-tests" ]; then
  timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
  tail -5 $OUT/pytest_gpu.log
fi
# the default run: the headline (C3) and, after it, the other single-GPU configs as short legs of the same process (config.other_configs_timed)
timeout 900 python bench.py --steps 20 --warmup 5 2> $OUT/bench_default.err | tail -1 > $OUT/bench_default.json
cut -c1-400 $OUT/bench_default.json
bash profiles/pmc_collect.sh $TAG C3 C2 C5 C1 > $OUT/pmc.log 2>&1
tail -4 $OUT/pmc.log | cut -c1-600
