#!/bin/bash
# the unguided headline workload on wg2 with the shared tail (VSPG_WG_SCHED=2) against wg (VSPG_WG_SCHED=1, the default)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03za
mkdir -p $OUT; : > $OUT/sweep.txt
cd $REPO
for rep in 1 2; do
for c in "1 8" "2 0" "2 8" "2 16" "2 32" "2 64"; do
  set -- $c
  VSPG_WG_SCHED=$1 VSPG_WG2_TAIL=$2 timeout -k 10 300 python bench.py --steps 32 --warmup 4 --no-cpu-baseline --no-pmc --no-reference-defaults 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('rep$rep sched=$1 tail=$2 %s kernel_ms %.4f value %.0f generic %s' % (d['roofline']['kernel'][:34], d['roofline']['kernel_ms'], d['value'], d.get('generic_instantiation',{}).get('value')))" | tee -a $OUT/sweep.txt
done; done
