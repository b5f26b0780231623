#!/bin/bash
# round 4, run m: NDS kernel chaining segments on boundary scenes: tests, then NDS timing of the cloud scene (VSPG_LIB=base for the A/B)
mkdir -p gpurun_out/r04
O=gpurun_out/r04
step() { local t=$1 log=$2; shift 2; timeout -k 10 $t "$@" > $O/$log 2>&1; local rc=$?; echo "$log rc=$rc"; tail -${TAILN:-3} $O/$log | cut -c1-600; if [ $rc -eq 124 ] || [ $rc -eq 137 ] || [ $rc -eq 134 ]; then exit $rc; fi; }
TAILN=6 step 900 m_tests.log python -m pytest tests -m gpu -q -x -k "boundaries or nds or temperature or cloud"
TAILN=6 step 400 m_nds_new.log python scripts/nds_timing.py scene scene-nvdb
[ -f build/variants/base.so ] && VSPG_LIB=$PWD/build/variants/base.so TAILN=6 step 400 m_nds_base.log python scripts/nds_timing.py scene scene-nvdb
exit 0
