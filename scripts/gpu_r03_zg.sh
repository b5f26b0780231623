#!/bin/bash
# NDS / NDS+ on the pipeline (k_wf_segment_vertex): the whole GPU suite, then per-lane vs pipeline timing on the 256^3 cloud
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03zg
mkdir -p $OUT
cd $REPO
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -4 $OUT/tests.log | cut -c1-300
grep -q "failed\|error" $OUT/tests.log && exit 1
timeout -k 10 400 python scripts/nds_timing.py 2>&1 | grep nds | tee $OUT/nds.txt
