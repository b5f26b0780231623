#!/bin/bash
# training waves on the workgroup kernel: parity (samples == oracle multiset, update), timing
mkdir -p gpurun_out/r03i
O=gpurun_out/r03i
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_host_adapter.py -m gpu -q -x -k "guid or train or config5 or cache" > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -12 $O/tests.log | cut -c1-300
[ $rc -eq 124 ] && exit 124
for k in "" lane; do
  VSPG_KERNEL=$k timeout -k 10 300 python bench.py --workload fog-guided --steps 16 --warmup 2 --train-waves 16 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('${k:-default}', d['roofline']['kernel'], 'kernel_ms %.3f' % d['roofline']['kernel_ms'], 'training', d['training'])" | tee -a $O/ab.txt
done
exit 0
