#!/bin/bash
mkdir -p gpurun_out
for k in wg lane; do
  VSPG_KERNEL=$k python bench.py --workload fog-guided --steps 16 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$k', d['roofline']['kernel'], 'kernel_ms %.3f' % d['roofline']['kernel_ms'], 'train ms %.2f' % d['training']['ms_per_wave'])"
done


