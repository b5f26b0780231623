#!/bin/bash
# per-kernel time of the wavefront pipeline on the cloud workload (rocprofv3 kernel trace, own run)
set -o pipefail
mkdir -p gpurun_out
OUT=$PWD/gpurun_out/wf_prof_${1:-a}
rm -rf $OUT; mkdir -p $OUT
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py --no-pmc --workload ${2:-cloud} --steps 4 --warmup 1 --no-cpu-baseline > $OUT/bench.log 2>&1 || { echo "trace failed"; tail -5 $OUT/bench.log; exit 1; }
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
cut -c1-200 $f | head -20
