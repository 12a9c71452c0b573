#!/bin/bash
# Lone-frame profile (run through gpurun): kernel trace + two SQ counter passes around tools/lone_frames.py.   $1 = tag, $2 = config (C3), rest: env assignments
TAG=${1:-x}; CFG=${2:-C3}; shift; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/frame_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=16
for kv in "$@"; do export "$kv"; done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 $REPO/tools/lone_frames.py --config $CFG > $OUT/trace.log 2>&1
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE" "SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM_RD"; do
  name=$(echo $grp | cut -d' ' -f1-2 | tr ' ' '_')
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/pmc_$name -o p -- python3 $REPO/tools/lone_frames.py --config $CFG > $OUT/pmc_$name.log 2>&1
done
python3 $REPO/profiles/frame_fold.py $OUT
