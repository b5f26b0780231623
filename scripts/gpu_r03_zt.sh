#!/bin/bash
# packed walk results for grey media: parity, then A/B against the full layout (VSPG_WF_COMPACT=0)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03zt
mkdir -p $OUT; : > $OUT/ab.txt
cd $REPO
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_nvdb_reader.py -x -q -m gpu -k "grid or nvdb or cloud or wavefront or placed or config5 or nds" > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -3 $OUT/tests.log
grep -q "failed\|error" $OUT/tests.log && exit 1
for rep in 1 2; do
for wl in cloud cloud-nvdb cloud-guided; do
for c in 1 0; do
  VSPG_WF_COMPACT=$c timeout -k 10 300 python bench.py --workload $wl --steps 8 --warmup 2 --train-waves 4 --no-cpu-baseline --no-pmc 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('rep$rep $wl compact=$c ms %.3f value %.1f' % (d['ms_per_step'], d['value']))" | tee -a $OUT/ab.txt
done; done; done
