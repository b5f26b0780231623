#!/bin/bash
# round 4: the new boundary tests first (fail fast), then the whole GPU suite
mkdir -p gpurun_out/r04
O=gpurun_out/r04
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "medium_boundaries" > $O/b_boundaries.log 2>&1; rc=$?; echo "boundaries rc=$rc"; tail -25 $O/b_boundaries.log | cut -c1-400
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
timeout -k 10 1000 python -m pytest tests -m gpu -q --deselect tests/test_gpu_parity.py::test_medium_boundaries_vs_oracle > $O/b_all.log 2>&1; rc2=$?; echo "all rc=$rc2"; tail -30 $O/b_all.log | cut -c1-300
exit 0
