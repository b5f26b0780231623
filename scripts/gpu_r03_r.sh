#!/bin/bash
# batched distance walk: parity tests of the grid / NanoVDB pipelines, then same-box A/B of the queue depth and trigger
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03r
mkdir -p $OUT
cd $REPO
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "grid or nvdb or cloud or wavefront or placed or explosion or heterogeneous or pipeline" > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -5 $OUT/tests.log
grep -q "failed" $OUT/tests.log && exit 1
WORKLOADS="cloud cloud-nvdb" bash scripts/gpu_variants_wl.sh
cp gpurun_out/variants_wl.txt $OUT/
