#!/bin/bash
# Per-launch k_trace_nearest durations (ms, one line per library) for the default build and every variant .so
cd "$(dirname "$0")/.."
for L in cadrays_amd/libcadrays_hip.so cadrays_amd/variants/*.so; do
  [ -e "$L" ] || continue
  printf "%-12s" "$(basename $L .so | sed s/libcadrays_hip/base/)"
  CRH_LIB_PATH=$PWD/$L bash tools/per_bounce.sh 2>&1 | grep "k_trace_nearest" | awk '{printf "%s ", $2} END {print ""}'
done
