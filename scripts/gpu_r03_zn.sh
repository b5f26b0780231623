#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03zn
mkdir -p $OUT; : > $OUT/ab.txt
cd $REPO
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "guided and (grid or nvdb or cloud or wavefront or config5 or pipeline)" > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -3 $OUT/tests.log
grep -q "failed\|error" $OUT/tests.log && exit 1
for rep in 1 2; do
for wl in cloud-guided cloud-nvdb-guided; do
  timeout -k 10 300 python bench.py --workload $wl --steps 8 --warmup 2 --train-waves 8 --no-cpu-baseline --no-pmc 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('rep$rep $wl ms %.3f value %.1f training %.2f' % (d['ms_per_step'], d['value'], d['training']['ms_per_wave']))" | tee -a $OUT/ab.txt
done; done
