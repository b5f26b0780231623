#!/bin/bash
# 3840x2160 (configs 2 / 4's resolution): one bench line per workload
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03zj
mkdir -p $OUT
cd $REPO
for wl in fog fog-guided cloud cloud-nvdb cloud-guided; do
  steps=16; case $wl in cloud*) steps=4;; esac
  timeout -k 10 500 python bench.py --workload $wl --xres 3840 --yres 2160 --steps $steps --warmup 2 --train-waves 8 --no-cpu-baseline --no-pmc --no-reference-defaults > $OUT/bench_4k_$wl.json 2> $OUT/bench_4k_$wl.err
  echo "$wl rc=$?"; tail -1 $OUT/bench_4k_$wl.json | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('  ms %.3f value %.1f' % (d['ms_per_step'], d['value']), d.get('training'))"
done
