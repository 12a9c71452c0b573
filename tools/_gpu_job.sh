for lib in cadrays_amd/libcadrays_hip.so cadrays_amd/variants/pf16.so cadrays_amd/variants/pf64.so; do
  echo "== $lib"
  CRH_LIB_PATH=$PWD/$lib CRH_LANES=1 python tools/bench_interactive.py 2>/dev/null | tail -1
  CRH_LIB_PATH=$PWD/$lib CRH_LANES=2 python tools/bench_interactive.py 2>/dev/null | tail -1
  CRH_LIB_PATH=$PWD/$lib python bench.py --no-cpu --no-interactive --steps 3 2>/dev/null | tail -1 | cut -c1-120
done
timeout 600 env CRH_LIB_PATH=$PWD/cadrays_amd/variants/pf64.so python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_two_level.py -m gpu -q 2>&1 | tail -2
