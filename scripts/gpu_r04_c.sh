#!/bin/bash
# round 4: boundary tests on the pipeline + everything that runs on the pipeline
mkdir -p gpurun_out/r04
O=gpurun_out/r04
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -s -k "medium_boundaries" > $O/c_boundaries.log 2>&1; rc=$?; echo "boundaries rc=$rc"; grep -v "^$" $O/c_boundaries.log | tail -45 | cut -c1-300
if [ $rc -ne 0 ]; then exit 0; fi
timeout -k 10 900 python -m pytest tests -m gpu -q --deselect tests/test_gpu_parity.py::test_medium_boundaries_vs_oracle > $O/c_all.log 2>&1; rc2=$?; echo "all rc=$rc2"; tail -12 $O/c_all.log | cut -c1-300
exit 0
