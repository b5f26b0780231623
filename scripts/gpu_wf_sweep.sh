#!/bin/bash
# sweep the walk kernels' knobs on the cloud workloads (kernel ms from the bench's HIP events)
mkdir -p gpurun_out; : > gpurun_out/wf_sweep.txt
for wl in cloud cloud-nvdb; do
for rounds in 1 2 3 4; do for refill in 8 16 24 32; do
  ms=$(VSPG_WF_ROUNDS=$rounds VSPG_WF_REFILL=$refill timeout -k 10 120 python bench.py --workload $wl --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; print('%.3f' % json.loads(sys.stdin.readlines()[-1])['roofline']['kernel_ms'])")
  echo "$wl rounds=$rounds refill=$refill ms=$ms" | tee -a gpurun_out/wf_sweep.txt
done; done; done
