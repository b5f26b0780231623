#!/bin/bash
# what the resampling routine's majorant pre-pass costs inside the dense kernels (timing-only build without it: wrong images)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03zm
mkdir -p $OUT
cd $REPO
for wl in cloud-guided cloud; do
for lib in "" build/variants/noprepass.so; do
  T=$OUT/trace_${wl}_$(basename ${lib:-default} .so); rm -rf $T; mkdir -p $T
  ( cd /tmp && export TMPDIR=/tmp && env ${lib:+VSPG_LIB=$REPO/$lib} timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $T -- python3 $REPO/bench.py --workload $wl --steps 4 --warmup 2 --train-waves 2 --no-cpu-baseline --no-pmc > $T/bench.log 2>&1 ) || { echo "trace failed"; tail -3 $T/bench.log; }
  f=$(find $T -name "*kernel_stats.csv" | head -1); echo "== $wl ${lib:-default}"; head -7 $f | cut -c1-150; tail -1 $T/bench.log | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('ms', d['ms_per_step'])"
done; done
