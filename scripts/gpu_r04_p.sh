#!/bin/bash
# round 4, run p: interface skips chained inside wf_segment_begin (start / begin / vertex kernels): tests, then A/B against build/variants/base.so
mkdir -p gpurun_out/r04
O=gpurun_out/r04
step() { local t=$1 log=$2; shift 2; timeout -k 10 $t "$@" > $O/$log 2>&1; local rc=$?; echo "$log rc=$rc"; tail -${TAILN:-3} $O/$log | cut -c1-600; if [ $rc -eq 124 ] || [ $rc -eq 137 ] || [ $rc -eq 134 ]; then exit $rc; fi; }
TAILN=6 step 900 p_tests.log python -m pytest tests -m gpu -q -x -k "boundaries or full_size_guided_cloud or cloud or temperature or scene_file"
WORKLOADS="cloud-scene cloud-scene-nvdb cloud-scene-guided" TAILN=20 step 900 p_variants_wl.log scripts/gpu_variants_wl.sh
cp gpurun_out/variants_wl.txt $O/p_variants_wl.txt
exit 0
