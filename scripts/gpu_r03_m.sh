#!/bin/bash
mkdir -p gpurun_out/r03m
O=gpurun_out/r03m
for dn in 0 1; do
VSPG_DENSE_BRICKS=$dn timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "grid or cloud or nvdb or transformed or emissive or config5" > $O/tests_$dn.log 2>&1; rc=$?; echo "dense=$dn tests rc=$rc"; tail -3 $O/tests_$dn.log | cut -c1-300
[ $rc -eq 124 ] && exit 124
for wl in cloud cloud-nvdb; do
VSPG_DENSE_BRICKS=$dn timeout -k 10 300 python bench.py --workload $wl --steps 6 --warmup 2 --no-cpu-baseline --no-pmc 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('dense=$dn $wl', 'ms %.2f' % d['ms_per_step'], 'value %.1f' % d['value'])" | tee -a $O/ab.txt
done
done
exit 0
