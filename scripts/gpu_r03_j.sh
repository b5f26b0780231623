#!/bin/bash
mkdir -p gpurun_out/r03j
O=gpurun_out/r03j
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_host_adapter.py -m gpu -q -x -k "guid or train or config5 or cache" > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -12 $O/tests.log | cut -c1-300
[ $rc -eq 124 ] && exit 124
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03j/trace; rm -rf $OUT; mkdir -p $OUT
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $REPO/bench.py --workload fog-guided --steps 8 --warmup 2 --train-waves 32 --no-cpu-baseline > $OUT/bench.log 2>&1 ) || { echo "trace failed"; tail -3 $OUT/bench.log; }
f=$(find $OUT -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_fog-guided.csv
tail -1 $OUT/bench.log | cut -c1-600
exit 0
