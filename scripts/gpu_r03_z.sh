#!/bin/bash
# wg2's shared tail (tiles handed out from a global head once a workgroup has started its static share): tests, then a sweep of
# the tail's share of the frame (VSPG_WG2_TAIL, in 64ths) on the reference-default guided workload (trained and training waves)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03z
mkdir -p $OUT; : > $OUT/sweep.txt
cd $REPO
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_host_adapter.py -x -q -m gpu -k "guided or guiding or train or scheduler or workgroup or full_size or scene_file" > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -3 $OUT/tests.log
grep -q "failed\|error" $OUT/tests.log && exit 1
for rep in 1 2; do
for t in 0 4 8 16 32 64; do
  VSPG_WG2_TAIL=$t timeout -k 10 300 python bench.py --workload fog-guided --steps 16 --warmup 2 --train-waves 24 --no-cpu-baseline --no-generic --no-pmc 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('rep$rep tail=$t trained %.4f ms  training %.3f ms' % (d['ms_per_step'], d['training']['ms_per_wave']))" | tee -a $OUT/sweep.txt
done; done
