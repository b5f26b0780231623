#!/bin/bash
# CPU test suite with the oracle and the host adapter built under AddressSanitizer + UBSan.
# GPU ASan is not available on the pool: sanitizers cover the CPU side only.
set -eo pipefail
cd "$(dirname "$0")/.."
make -C oracle asan
make -C vspg-pbrt-v4_amd/host asan
echo "== host_selftest under ASan/UBSan"
ASAN_OPTIONS=detect_leaks=1 ./vspg-pbrt-v4_amd/host/host_selftest_asan
echo "== pytest -m 'not gpu' with liboracle_asan.so"
LIBASAN=$(gcc -print-file-name=libasan.so)
# python itself is not instrumented: preload the runtime, leak checking off (the interpreter never frees everything)
LD_PRELOAD=$LIBASAN ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 VSPG_ORACLE_SO=liboracle_asan.so OMP_NUM_THREADS=4 \
  python -m pytest tests -x -q -s -m "not gpu" -p no:cacheprovider "$@" 2>&1 | grep -v "^\[Gloo\]"
