#!/bin/bash
# resident workgroups per CU of the two walk kernels (they run side by side since the fused pipeline): sweep
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03v
mkdir -p $OUT; : > $OUT/sweep.txt
cd $REPO
for rep in 1 2; do
for wl in cloud cloud-nvdb; do
for c in "4 5" "4 4" "4 3" "4 2" "3 3" "3 2" "3 1" "2 2"; do
  set -- $c
  ms=$(VSPG_WF_WALK_BLOCKS=$1 VSPG_WF_SHADOW_BLOCKS=$2 timeout -k 10 200 python bench.py --workload $wl --steps 8 --warmup 2 --no-cpu-baseline --no-generic --no-pmc 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%.3f' % d['ms_per_step'])")
  echo "rep$rep $wl walk=$1 shadow=$2 $ms" | tee -a $OUT/sweep.txt
done; done; done
