#!/bin/bash
# measurement of record: kernel-trace stats + PMC passes for the bench workloads (own runs, --pmc never mixed with trace domains)
TAG=${1:-r05}
PART=${2:-all}   # trace | pmc | all (the two parts fit one gpurun call each)
mkdir -p gpurun_out/final
REPO=${GRAFT_REPO_ROOT:-/root/repo}
[ $PART = pmc ] || for wl in ${TRACE_WL:-fog fog-guided cloud cloud-nvdb cloud-guided cloud-scene cloud-scene-nvdb cloud-scene-guided}; do
  OUT=$REPO/gpurun_out/final/trace_$wl; rm -rf $OUT; mkdir -p $OUT
  steps=16; case $wl in cloud*) steps=4;; esac
  ( cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $REPO/bench.py --workload $wl --steps $steps --warmup 2 --train-waves 4 --no-cpu-baseline --no-pmc > $OUT/bench.log 2>&1 ) || { echo "trace $wl failed"; tail -3 $OUT/bench.log; }
  f=$(find $OUT -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/final/${TAG}_kernel_stats_${wl}.csv && head -6 $f | cut -c1-160
  tail -1 $OUT/bench.log > gpurun_out/final/${TAG}_bench_under_trace_${wl}.json
done
[ $PART = trace ] || for wl in ${PMC_WL:-fog fog-guided cloud cloud-nvdb cloud-guided cloud-scene}; do
  bash scripts/gpu_wf_pmc.sh final_$wl $wl > gpurun_out/final/pmc_$wl.log 2>&1
  python3 scripts/summarize_pmc.py gpurun_out/pmc_final_$wl gpurun_out/final/${TAG}_pmc_${wl}.json $wl 3 > gpurun_out/final/${TAG}_pmc_${wl}.txt
  cat gpurun_out/final/${TAG}_pmc_${wl}.txt
done
