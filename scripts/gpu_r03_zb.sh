#!/bin/bash
# wg2 as the default scheduler (global tile head, parked samples resolved by the next launch): the whole GPU suite, then
# A/B of the deferred resolve on the headline and the reference-default workloads
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03zb
mkdir -p $OUT; : > $OUT/ab.txt
cd $REPO
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -4 $OUT/tests.log | cut -c1-300
grep -q "failed\|error" $OUT/tests.log && exit 1
for rep in 1 2; do
for d in 1 0; do
  VSPG_WG2_DEFER=$d timeout -k 10 300 python bench.py --steps 32 --warmup 4 --no-cpu-baseline --no-pmc 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); rd=d['reference_defaults']; print('rep$rep defer=$d %s ms %.4f value %.0f generic %.0f | guided trained %.4f training %.3f' % (d['roofline']['kernel'][:40], d['ms_per_step'], d['value'], d['generic_instantiation']['value'], rd['ms_per_trained_wave'], rd['training']['ms_per_wave']))" | tee -a $OUT/ab.txt
done; done
