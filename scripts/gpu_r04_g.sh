#!/bin/bash
# round 4: after the BND-by-instantiation split -- pipeline tests, scene-file test, A/B against the no-boundary-code build, per-phase profile of the headline kernel
mkdir -p gpurun_out/r04
O=gpurun_out/r04
step() { local t=$1 log=$2; shift 2; timeout -k 10 $t "$@" > $O/$log 2>&1; local rc=$?; echo "$log rc=$rc"; tail -${TAILN:-3} $O/$log | cut -c1-600; if [ $rc -eq 124 ] || [ $rc -eq 137 ] || [ $rc -eq 134 ]; then exit $rc; fi; }
TAILN=6 step 400 g_scenefile.log python -m pytest tests/test_host_adapter.py -m gpu -q -x -k "boundaries"
TAILN=6 step 600 g_tests.log python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "boundaries or cloud or grid or nvdb or nds or wavefront or shard or config5"
WORKLOADS="cloud cloud-nvdb cloud-guided" TAILN=14 step 900 g_variants.log scripts/gpu_variants_wl.sh
cp gpurun_out/variants_wl.txt $O/g_variants_wl.txt
TAILN=40 step 300 g_prof_fog.txt python scripts/prof_sections.py
exit 0
