#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03zp
mkdir -p $OUT
cd $REPO
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "nds or NDS or tr_buffer or emissive" > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -3 $OUT/tests.log
grep -q "failed\|error" $OUT/tests.log && exit 1
timeout -k 10 400 python scripts/nds_timing.py 2>&1 | grep nds | tee $OUT/nds.txt
