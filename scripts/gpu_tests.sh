#!/bin/bash
# the whole GPU suite (and nothing else)
mkdir -p gpurun_out/tests
timeout -k 10 1100 python -m pytest tests -m gpu -q -x > gpurun_out/tests/tests_gpu.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -14 gpurun_out/tests/tests_gpu.log | cut -c1-300
exit $rc
