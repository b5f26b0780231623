#!/bin/bash
# compact pool records + guided wg2 as default: the whole GPU suite, then timings
mkdir -p gpurun_out/r03f
O=gpurun_out/r03f
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $O/tests_gpu.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -12 $O/tests_gpu.log | cut -c1-300
[ $rc -eq 124 ] && exit 124
for rep in 1 2; do for sch in 1 2; do
  VSPG_WG_SCHED=$sch timeout -k 10 200 python bench.py --steps 48 --warmup 4 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('sched $sch fog  kernel_ms %.4f value %.0f generic %s' % (d['roofline']['kernel_ms'], d['value'], d.get('generic_instantiation',{}).get('value')))" | tee -a $O/ab.txt
done; done
for k in "" lane; do for ng in "" 1; do
  VSPG_NO_GREY_GUIDED=$ng VSPG_KERNEL=$k timeout -k 10 300 python bench.py --workload fog-guided --steps 16 --warmup 2 --train-waves 8 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('${k:-default} nogrey=$ng', d['roofline']['kernel'], 'kernel_ms %.3f' % d['roofline']['kernel_ms'], 'train ms %.2f' % d['training']['ms_per_wave'])" | tee -a $O/ab.txt
done; done
exit 0
