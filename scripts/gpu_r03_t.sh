#!/bin/bash
# guided pipeline: A/B of the vertex kernel's shapes (same box)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03t
mkdir -p $OUT
cd $REPO
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "guided and (grid or nvdb or cloud or pipeline or config5)" > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -3 $OUT/tests.log
grep -q "failed\|error" $OUT/tests.log && exit 1
WORKLOADS="cloud-guided cloud-nvdb-guided" bash scripts/gpu_variants_wl.sh
cp gpurun_out/variants_wl.txt $OUT/
