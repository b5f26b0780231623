#!/bin/bash
# after the group-width experiments: the pipeline's tests again (incl. the concurrent-vs-single-stream test), one cloud bench line
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03y
mkdir -p $OUT
cd $REPO
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "grid or nvdb or cloud or wavefront or placed or config5" > $OUT/tests.log 2>&1
echo "tests rc=$?"; tail -3 $OUT/tests.log
grep -q "failed\|error" $OUT/tests.log && exit 1
for wl in cloud cloud-guided; do
timeout -k 10 300 python bench.py --workload $wl --steps 8 --warmup 2 --no-cpu-baseline --no-pmc 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$wl', 'ms %.2f' % d['ms_per_step'], 'value %.1f' % d['value'])"
done
