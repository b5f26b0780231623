#!/bin/bash
mkdir -p gpurun_out/r03l
O=gpurun_out/r03l
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > $O/tests_gpu.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -14 $O/tests_gpu.log | cut -c1-300
[ $rc -eq 124 ] && exit 124
for wl in cloud cloud-nvdb; do
timeout -k 10 300 python bench.py --workload $wl --steps 6 --warmup 2 --no-cpu-baseline --no-pmc 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$wl', 'ms %.2f' % d['ms_per_step'], 'value %.1f' % d['value'])" | tee -a $O/ab.txt
done
exit 0
