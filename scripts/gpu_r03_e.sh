#!/bin/bash
mkdir -p gpurun_out/r03e
O=gpurun_out/r03e
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "guided" > $O/tests_guided.log 2>&1; rc=$?; echo "guided tests rc=$rc"; tail -4 $O/tests_guided.log | cut -c1-300
[ $rc -eq 124 ] && exit 124
for lib in "" build/variants/g128.so build/variants/g128b.so; do
  VSPG_LIB=${lib:+$PWD/$lib} timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "guided_workgroup" > $O/tests_var.log 2>&1; echo "variant ${lib:-default} parity rc=$?"; tail -2 $O/tests_var.log | cut -c1-200
  for k in wg lane; do
  VSPG_LIB=${lib:+$PWD/$lib} VSPG_KERNEL=$k timeout -k 10 300 python bench.py --workload fog-guided --steps 16 --warmup 2 --train-waves 8 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('${lib:-default} $k', d['roofline']['kernel'], 'kernel_ms %.3f' % d['roofline']['kernel_ms'], 'train ms %.2f' % d['training']['ms_per_wave'])" | tee -a $O/ab.txt
  done
done
exit 0
