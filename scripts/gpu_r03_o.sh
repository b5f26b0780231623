#!/bin/bash
mkdir -p gpurun_out/r03o
O=gpurun_out/r03o
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "full_size_cloud or grid or cloud or nvdb" > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $O/tests.log | cut -c1-300
[ $rc -eq 124 ] && exit 124
for ns in 1 2 3 4; do
for wl in cloud cloud-nvdb; do
VSPG_WF_SLICES=$ns timeout -k 10 300 python bench.py --workload $wl --steps 6 --warmup 2 --no-cpu-baseline --no-pmc 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('slices=$ns $wl', 'ms %.2f' % d['ms_per_step'], 'value %.1f' % d['value'])" | tee -a $O/ab.txt
done
done
exit 0
