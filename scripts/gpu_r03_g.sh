#!/bin/bash
# guided wg2 with prefetched region loads: parity, then timing of pool-size / kd-staging / scheduler-flag variants
mkdir -p gpurun_out/r03g
O=gpurun_out/r03g
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "guid or train or config5" > $O/tests_guided.log 2>&1; rc=$?; echo "guided tests rc=$rc"; tail -4 $O/tests_guided.log | cut -c1-300
[ $rc -eq 124 ] && exit 124
for lib in "" build/variants/np384_448.so build/variants/kd256.so build/variants/noilp.so; do
  for ng in 0 1; do
  export VSPG_LIB=${lib:+$PWD/$lib}; [ -z "$lib" ] && unset VSPG_LIB
  if [ $ng = 1 ]; then export VSPG_NO_GREY_GUIDED=1; else unset VSPG_NO_GREY_GUIDED; fi
  timeout -k 10 300 python bench.py --workload fog-guided --steps 16 --warmup 2 --train-waves 8 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('${lib:-default} nogrey=$ng', d['roofline']['kernel'], 'kernel_ms %.3f' % d['roofline']['kernel_ms'], 'train ms %.2f' % d['training']['ms_per_wave'])" | tee -a $O/ab.txt
  done
done
unset VSPG_LIB VSPG_NO_GREY_GUIDED
VSPG_KERNEL=lane timeout -k 10 300 python bench.py --workload fog-guided --steps 16 --warmup 2 --train-waves 8 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('lane', d['roofline']['kernel'], 'kernel_ms %.3f' % d['roofline']['kernel_ms'], 'train ms %.2f' % d['training']['ms_per_wave'])" | tee -a $O/ab.txt
for sch in 1 2; do
  VSPG_WG_SCHED=$sch timeout -k 10 200 python bench.py --steps 48 --warmup 4 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('sched $sch fog  kernel_ms %.4f value %.0f generic %s' % (d['roofline']['kernel_ms'], d['value'], d.get('generic_instantiation',{}).get('value')))" | tee -a $O/ab.txt
done
exit 0
