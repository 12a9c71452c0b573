#!/bin/bash
# On the GPU box: bench every variant .so (and the default build as "base"); one line per run into gpurun_out/ab.txt
#   tools/ab_run.sh [bench args...]
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
: > gpurun_out/ab.txt
run() {
  local name=$1 lib=$2; shift 2
  CRH_LIB_PATH=$lib python bench.py --no-cpu --other-configs none --steps 3 --warmup 1 "$@" 2>/dev/null | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); r = j['roofline']
    print('%-28s %9.1f Mrays/s  %8.3f ms/step  trace %.3f ms  share %.3f' % ('$name', j['value'], j['ms_per_step'], r['avg_launch_ms'], r['kernel_time_share']))
" >> gpurun_out/ab.txt
}
run base cadrays_amd/libcadrays_hip.so "$@"
for f in cadrays_amd/variants/*.so; do [ -e "$f" ] && run $(basename $f .so) $f "$@"; done
run base2 cadrays_amd/libcadrays_hip.so "$@"
cat gpurun_out/ab.txt
