#!/bin/bash
# round 4, run k: A/B of workgroup-kernel scheduler changes (library variants under build/variants/ against the default), tests first
mkdir -p gpurun_out/r04
O=gpurun_out/r04
step() { local t=$1 log=$2; shift 2; timeout -k 10 $t "$@" > $O/$log 2>&1; local rc=$?; echo "$log rc=$rc"; tail -${TAILN:-3} $O/$log | cut -c1-600; if [ $rc -eq 124 ] || [ $rc -eq 137 ] || [ $rc -eq 134 ]; then exit $rc; fi; }
TAILN=6 step 600 k_tests.log python -m pytest tests -m gpu -q -x -k "workgroup or full_size_wave or full_size_guided_wave or schedulers or shard or triangle or boundaries or parity_small or film"
TAILN=12 step 500 k_variants.log scripts/gpu_variants.sh
cp gpurun_out/variants.txt $O/k_variants.txt
WORKLOADS="fog-guided" TAILN=8 step 500 k_variants_wl.log scripts/gpu_variants_wl.sh
cp gpurun_out/variants_wl.txt $O/k_variants_wl.txt
exit 0
