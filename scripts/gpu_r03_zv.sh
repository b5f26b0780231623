#!/bin/bash
# NanoVDB semantics: the next majorant cell requested one step early (default) against not (build/variants/nopref.so)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03zv
mkdir -p $OUT
cd $REPO
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_nvdb_reader.py -x -q -m gpu -k "nvdb or config5 or explosion or placed" > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -3 $OUT/tests.log
grep -q "failed\|error" $OUT/tests.log && exit 1
WORKLOADS="cloud-nvdb cloud-nvdb-guided" bash scripts/gpu_variants_wl.sh
cp gpurun_out/variants_wl.txt $OUT/
