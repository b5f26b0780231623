#!/bin/bash
# round 4, run j: blackbody emission of temperature grids under "nds" -- the whole GPU suite (new tests included)
mkdir -p gpurun_out/r04
O=gpurun_out/r04
step() { local t=$1 log=$2; shift 2; timeout -k 10 $t "$@" > $O/$log 2>&1; local rc=$?; echo "$log rc=$rc"; tail -${TAILN:-3} $O/$log | cut -c1-600; if [ $rc -eq 124 ] || [ $rc -eq 137 ] || [ $rc -eq 134 ]; then exit $rc; fi; }
TAILN=12 step 1100 j_tests_all.log python -m pytest tests -m gpu -q -x -s
grep -h "emission adds" $O/j_tests_all.log
exit 0
