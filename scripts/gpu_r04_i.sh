#!/bin/bash
mkdir -p gpurun_out/r04
O=gpurun_out/r04
step() { local t=$1 log=$2; shift 2; timeout -k 10 $t "$@" > $O/$log 2>&1; local rc=$?; echo "$log rc=$rc"; tail -${TAILN:-3} $O/$log | cut -c1-600; if [ $rc -eq 124 ] || [ $rc -eq 137 ] || [ $rc -eq 134 ]; then exit $rc; fi; }
TAILN=8 step 600 i_tests.log python -m pytest tests -m gpu -q -x -s -k "full_size_guided_cloud or config1 or medium_boundaries"
WORKLOADS="cloud cloud-guided cloud-scene" TAILN=26 step 1000 i_variants.log scripts/gpu_variants_wl.sh
cp gpurun_out/variants_wl.txt $O/i_variants_wl.txt
exit 0
