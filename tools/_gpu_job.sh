for i in 1 2; do
echo "== current lib, run $i"
timeout 900 python -m pytest tests/test_compare_runs.py tests/test_gpu_parity.py -m gpu -q -k "preview or crh_reduce or lookahead" 2>&1 | grep -E "^FAILED|passed|failed|AssertionError:" | cut -c1-400
echo "== r1 lib, run $i"
CRH_LIB_PATH=$PWD/cadrays_amd/variants/r1.so timeout 900 python -m pytest tests/test_compare_runs.py tests/test_gpu_parity.py -m gpu -q -k "preview or crh_reduce or lookahead" 2>&1 | grep -E "^FAILED|passed|failed|AssertionError:" | cut -c1-400
done
