#!/bin/bash
# two ranks on one card (gloo stand-in for RCCL, VSPG_BENCH_REHEARSE=1): the multi-GPU bench path end to end
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03zh
mkdir -p $OUT
cd $REPO
for wl in fog fog-guided cloud; do
  VSPG_BENCH_REHEARSE=1 timeout -k 10 500 python bench.py --gpus 2 --workload $wl --steps 8 --warmup 2 --no-cpu-baseline --no-pmc > $OUT/rehearse_$wl.json 2> $OUT/rehearse_$wl.err
  echo "$wl rc=$?"; tail -1 $OUT/rehearse_$wl.json | cut -c1-400; tail -2 $OUT/rehearse_$wl.err | cut -c1-200
done
