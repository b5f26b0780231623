#!/bin/bash
# runtime record-group width vs the compile-time builds (g4 / g8), same box; then NDS on the cloud: per-lane vs workgroup kernel
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03x
mkdir -p $OUT
cd $REPO
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "grid or nvdb or cloud or wavefront or placed or config5" > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -3 $OUT/tests.log
grep -q "failed\|error" $OUT/tests.log && exit 1
WORKLOADS="cloud cloud-nvdb cloud-guided cloud-nvdb-guided" bash scripts/gpu_variants_wl.sh
cp gpurun_out/variants_wl.txt $OUT/
timeout -k 10 400 python scripts/nds_timing.py 2>&1 | tee $OUT/nds.txt
