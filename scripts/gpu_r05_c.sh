#!/bin/bash
# round 5: scheduler tests, then same-box A/B of the library builds under build/variants/ on the headline workload
mkdir -p gpurun_out/r05
timeout -k 10 240 python -m pytest tests/test_gpu_parity.py -q -x -k "schedulers_are_bit_identical or parked_samples" > gpurun_out/r05/c_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r05/c_tests.log
[ $rc -eq 0 ] || exit $rc
VARIANT_ARGS="--no-pmc --no-reference-defaults" bash scripts/gpu_variants.sh
cp gpurun_out/variants.txt gpurun_out/r05/c_variants.txt
