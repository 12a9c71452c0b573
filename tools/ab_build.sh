#!/bin/bash
# A/B builds of libcadrays_hip.so with extra -D flags:  tools/ab_build.sh NAME "-DCRH_X=1 ..." [NAME2 "flags2" ...]
# Outputs cadrays_amd/variants/NAME.so (git-ignored; they travel to the GPU box).  Run with CRH_LIB_PATH=... python bench.py
set -e
cd "$(dirname "$0")/../cadrays_amd/csrc"
mkdir -p ../variants
BASE="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fvisibility=hidden -Wno-unused-function -Wno-unused-result -Wno-unused-value -Wno-inline-asm"
make -s crh_context.o crh_scene.o crh_schedule.o crh_readback.o crh_reduce.o crh_debug.o bvh_builder.o
while [ $# -ge 2 ]; do
  NAME=$1; FL=$2; shift 2
  ( /opt/rocm/bin/hipcc $BASE $FL -c kernels.hip -o ../variants/$NAME.kernels.o 2>/dev/null &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants/$NAME.so ../variants/$NAME.kernels.o crh_context.o crh_scene.o crh_schedule.o crh_readback.o crh_reduce.o crh_debug.o bvh_builder.o -lpthread &&
    rm -f ../variants/$NAME.kernels.o && echo "built $NAME [$FL]" ) &
  while [ $(jobs -r | wc -l) -ge 6 ]; do sleep 0.5; done
done
wait
