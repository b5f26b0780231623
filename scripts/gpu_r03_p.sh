#!/bin/bash
# fused-field training update: its tests, then the per-kernel profile of the training waves
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03p
mkdir -p $OUT
cd $REPO
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "training or train or guiding or guided" > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -5 $OUT/tests.log
grep -q "failed" $OUT/tests.log && exit 1
bash scripts/gpu_trainprof.sh > $OUT/trainprof.txt 2>&1
tail -45 $OUT/trainprof.txt
