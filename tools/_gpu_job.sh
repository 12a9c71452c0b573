echo "== donate parity"
CRH_DONATE=1 timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | grep -E "passed|failed|FAILED" | head -5
CRH_DONATE=1 timeout 900 python -m pytest tests/test_gpu_fuzz.py tests/test_two_level.py tests/test_adaptive.py tests/test_textures.py tests/test_golden.py tests/test_gpu_kat.py -m gpu -x -q 2>&1 | grep -E "passed|failed|FAILED" | head -5
for dn in 0 1; do echo "== CRH_DONATE=$dn"; CRH_DONATE=$dn timeout 300 python tools/bench_interactive.py 2>/dev/null | tail -1; done
cd /tmp && export TMPDIR=/tmp
CRH_DONATE=1 CRH_LANES=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2k/trace1 -o t -- python3 $GRAFT_REPO_ROOT/tools/bench_interactive.py --frames 32 > /dev/null 2>&1
python3 $GRAFT_REPO_ROOT/tools/frame_timeline.py $(find $GRAFT_REPO_ROOT/gpurun_out/r2k/trace1 -name "*kernel_trace.csv")
