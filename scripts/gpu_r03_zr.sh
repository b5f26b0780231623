#!/bin/bash
# per-lane kernel: static share of the frame per wavefront (64ths) on the triangle scenes
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03zr
mkdir -p $OUT; : > $OUT/sweep.txt
cd $REPO
for t in 48 32 16 0; do
  VSPG_LANE_STATIC=$t timeout -k 10 400 python scripts/tri_timing.py 2>/dev/null | grep "fog" | sed "s/^/static=$t /" | tee -a $OUT/sweep.txt
done
