#!/bin/bash
# Round 6: why is a lone frame 0.2 ms dearer from the Python host than from the C++ host (same calls)?  The Python process runs on the HIP runtime bundled with
# the torch wheel (ROCm 7.0), the C++ host on /opt/rocm's (7.2); and both wait in hipStreamSynchronize, whose wake-up the runtime's environment knobs change.
#   gpurun -- 'bash tools/host_runtime_probe.sh [C3]'   -> gpurun_out/host_probe/
CFG=${1:-C3}
OUT=gpurun_out/host_probe; mkdir -p $OUT
python - <<PY
import sys; sys.path.insert(0, '.')
from cadrays_amd import scenes, scene_io
scene_io.save_scene('/tmp/$CFG.crhscene', scenes.baseline_config('$CFG'))
PY
run() {  # $1 = label, rest = env assignments
  label=$1; shift
  echo "== $label" | tee -a $OUT/$CFG.txt
  env "$@" python tools/bench_redraw.py --config $CFG --trials 25 2>&1 | tail -1 | cut -c1-700 >> $OUT/$CFG.txt
  env "$@" python tools/bench_redraw.py --config $CFG --trials 25 --no-torch 2>&1 | tail -1 | cut -c1-700 >> $OUT/$CFG.txt
  for loop in lone drag display; do
    n=384; [ $loop = lone ] && n=25
    env "$@" cadrays_amd/host/cadrays_headless /tmp/$CFG.crhscene $n --loop $loop 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('  c++ host', d['loop'], d['lone_frame_ms_median'] or d['loop_frames_per_s'])" >> $OUT/$CFG.txt
  done
}
run "default" X=1
run "ROC_ACTIVE_WAIT_TIMEOUT=10000" ROC_ACTIVE_WAIT_TIMEOUT=10000
run "HSA_ENABLE_INTERRUPT=0" HSA_ENABLE_INTERRUPT=0
run "ROC_CPU_WAIT_FOR_SIGNAL=1" ROC_CPU_WAIT_FOR_SIGNAL=1
run "default again" X=1
cat $OUT/$CFG.txt
