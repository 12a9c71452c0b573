#!/bin/bash
# Profile passes on the GPU box (run through gpurun).  $1 = tag.  Outputs under gpurun_out/prof_$1/.
# Kernel-trace stats and each PMC group are separate runs (rocprofv3 --pmc is never combined with tracing domains other than kernel-trace).
TAG=${1:-x}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 4 --warmup 1 --no-cpu ${BENCH_ARGS}"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 $REPO/bench.py $ARGS > $OUT/trace.log 2>&1
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM_RD" ; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/pmc_$name -o p -- python3 $REPO/bench.py $ARGS > $OUT/pmc_$name.log 2>&1
done
python3 $REPO/profiles/profile_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
