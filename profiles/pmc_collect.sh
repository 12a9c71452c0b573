#!/bin/bash
# Memory-side counter passes for bench.py's roofline.traffic, one set per config (run through gpurun):
#   ./profiles/pmc_collect.sh <tag> [configs...]        default configs: C3 C2 C5 C1
# Per config: one rocprofv3 --kernel-trace --stats pass and separate --pmc passes (FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum |
# TCC_REQ_sum TCC_READ_sum | two SQ groups; --pmc is only ever combined with --kernel-trace), all around the SAME command bench.py's
# default run uses.  profiles/pmc_fold.py then writes profiles/pmc_traffic.json, keyed by the hash of the kernel sources, so
# that bench.py can refuse figures that belong to another build.  Outputs under gpurun_out/pmc_<tag>/.
TAG=${1:-x}; shift
CFGS=${@:-C3 C2 C5 C1}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for cfg in $CFGS; do
  ARGS="--config $cfg --steps 2 --warmup 1 --no-cpu --no-interactive --no-parity --other-configs none"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$cfg/trace -o t -- python3 $REPO/bench.py $ARGS > $OUT/$cfg.trace.log 2>&1
  grep '^{"metric"' $OUT/$cfg.trace.log | tail -1 > $OUT/$cfg.bench.json
  # memory side (FETCH_SIZE costs 3 of the 4 TCC slots, WRITE_SIZE 2: separate passes), L2 hit / miss, L2-side requests, and one SQ pass
  # (8 slots): issue activity, instruction counts and lane utilisation of the dominant kernel -- the measured ceilings bench.py reports
  for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE"; do
    name=$(echo $grp | tr ' ' '_')
    rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/$cfg/pmc_$name -o p -- python3 $REPO/bench.py $ARGS > $OUT/$cfg.pmc_$name.log 2>&1
  done
done
python3 $REPO/profiles/pmc_fold.py $OUT $CFGS
python3 $REPO/profiles/pmc_per_bounce.py $OUT $CFGS > $OUT/per_bounce_counters.txt 2>&1
