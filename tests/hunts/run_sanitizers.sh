#!/bin/bash
# CPU-side sanitizer pass: the oracle (C, OpenMP) and the product's host-side BVH builder (C++ threads,
# atomics, futures) under AddressSanitizer, UndefinedBehaviorSanitizer and ThreadSanitizer.  Never on the GPU.
#   tests/hunts/run_sanitizers.sh [asan ubsan tsan]      -> sanitizers/<kind>.log (+ readers_malformed.log), exit status 1 on any report
cd "$(dirname "$0")/../.."
KINDS=${@:-asan ubsan tsan}
OUT=sanitizers; mkdir -p $OUT
rc=0
for k in $KINDS; do
  make -s -C oracle $k || { echo "$k: build failed"; rc=1; continue; }
  log=$OUT/$k.log; : > $log
  # 1. the multi-threaded builder
  case $k in
    tsan) env="TSAN_OPTIONS=halt_on_error=0:report_signal_unsafe=0" ;;
    asan) env="ASAN_OPTIONS=detect_leaks=1" ;;
    *)    env="UBSAN_OPTIONS=print_stacktrace=1" ;;
  esac
  env $env oracle/san/$k/sanitize_host 30000 >> $log 2>&1 || { echo "$k: sanitize_host failed"; rc=1; }
  # 2. the oracle through the CPU test-suite's own oracle tests (the sanitized library is preloaded in place of the -O2 one)
  if [ $k != tsan ]; then       # libgomp is not TSan-instrumented: its barriers read as races; the oracle's OpenMP loops only touch disjoint pixels
    pre=$(gcc -print-file-name=lib${k}.so)
    env $env LD_PRELOAD=$pre CRH_ORACLE_LIB=$PWD/oracle/san/$k/libcrh_oracle.so ASAN_OPTIONS=detect_leaks=0 \
      python -m pytest tests/test_oracle_kat.py tests/test_golden.py tests/test_node_quantiser.py tests/test_geometric_truth.py tests/test_adaptive.py tests/test_two_level.py tests/test_textures.py tests/test_visibility.py tests/test_cad_like.py tests/test_icon_features.py -x -q -m "not gpu" -k "not product_builder" >> $log 2>&1 \
      || { echo "$k: oracle tests failed"; rc=1; }
  fi
  if grep -E "ERROR: (Address|Thread|Leak)Sanitizer|runtime error:|WARNING: ThreadSanitizer" $log > /dev/null; then echo "$k: sanitizer reports in $log"; rc=1; else echo "$k: clean"; fi
done
# 3. the C++ file readers (model.tcl, PLY, PNG, JPEG: they parse files a user hands them): ASan + UBSan build, truncated / bit-flipped /
#    spliced inputs (tools/fuzz_readers_malformed.py); anything but a clean exit or an error message is a finding
if g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-sanitize-recover=undefined -o /tmp/model_tcl_dump_asan cadrays_amd/host/model_tcl_dump.cpp -lz; then
  python3 tools/fuzz_readers_malformed.py /tmp/model_tcl_dump_asan 300 1 > $OUT/readers_malformed.log 2>&1 && echo "readers: clean" || { echo "readers: findings in $OUT/readers_malformed.log"; rc=1; }
else echo "readers: build failed"; rc=1; fi
exit $rc
