#!/bin/bash
# kernel-trace A/B of the two workgroup schedulers (fog): where does the time go
mkdir -p gpurun_out/r03c
REPO=${GRAFT_REPO_ROOT:-/root/repo}
for sch in 1 2; do
  OUT=$REPO/gpurun_out/r03c/trace_$sch; rm -rf $OUT; mkdir -p $OUT
  ( cd /tmp && export TMPDIR=/tmp && VSPG_WG_SCHED=$sch timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $REPO/bench.py --steps 32 --warmup 4 --no-cpu-baseline --no-generic > $OUT/bench.log 2>&1 ) || { echo "trace failed"; tail -3 $OUT/bench.log; }
  f=$(find $OUT -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/r03c/kernel_stats_sched$sch.csv && head -5 $f | cut -c1-200
done
exit 0
