#!/bin/bash
# per-kernel times of the fused pipeline (cloud, cloud-guided) + A/B of the vertex kernel's launch bound
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03u
rm -rf $OUT; mkdir -p $OUT
cd $REPO
for wl in cloud cloud-guided; do
  T=$OUT/trace_$wl; mkdir -p $T
  ( cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $T -- python3 $REPO/bench.py --workload $wl --steps 4 --warmup 2 --train-waves 4 --no-cpu-baseline --no-pmc > $T/bench.log 2>&1 ) || { echo "trace $wl failed"; tail -3 $T/bench.log; exit 1; }
  f=$(find $T -name "*kernel_stats.csv" | head -1); cp $f $OUT/kernel_stats_$wl.csv; head -9 $f | cut -c1-200
done
WORKLOADS="cloud cloud-nvdb" bash scripts/gpu_variants_wl.sh
cp gpurun_out/variants_wl.txt $OUT/
