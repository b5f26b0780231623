#!/bin/bash
# record group width of the wavefront pipeline (16 B / 32 B / 64 B / 128 B per slot per group): parity subset, then same-box A/B
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03w
mkdir -p $OUT
cd $REPO
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "grid or nvdb or cloud or wavefront or placed or config5" > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -3 $OUT/tests.log
grep -q "failed\|error" $OUT/tests.log && exit 1
WORKLOADS="cloud cloud-nvdb cloud-guided" bash scripts/gpu_variants_wl.sh
cp gpurun_out/variants_wl.txt $OUT/
