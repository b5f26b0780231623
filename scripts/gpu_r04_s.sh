#!/bin/bash
# round 4, run s: the walks' per-cell exponential sample decided by the native log2 where that is safe: tests (full-size ones included), A/B
mkdir -p gpurun_out/r04
O=gpurun_out/r04
step() { local t=$1 log=$2; shift 2; timeout -k 10 $t "$@" > $O/$log 2>&1; local rc=$?; echo "$log rc=$rc"; tail -${TAILN:-3} $O/$log | cut -c1-900; if [ $rc -ne 0 ]; then exit $rc; fi; }
TAILN=25 step 1000 s_tests.log python -m pytest tests -m gpu -q -x -k "cloud or boundaries or temperature or scene_file or pipeline or nds or full_size or config5 or nvdb or grid"
WORKLOADS="cloud cloud-nvdb cloud-guided cloud-scene cloud-scene-nvdb" TAILN=24 step 900 s_variants_wl.log scripts/gpu_variants_wl.sh
cp gpurun_out/variants_wl.txt $O/s_variants_wl.txt
exit 0
