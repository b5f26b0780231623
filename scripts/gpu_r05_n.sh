#!/bin/bash
# round 5, step n: resident workgroups per CU of the merged walk kernel (VSPG_WF_MERGED_BLOCKS = 1..4 of 256 threads) on the workloads
# that run it: a tail-dominated kernel may want FEWER wavefronts (more jobs per lane, thin wavefronts that do not share a SIMD four ways)
mkdir -p gpurun_out/r05; : > gpurun_out/r05/merged_blocks.txt
for wl in cloud-scene cloud-scene-nvdb cloud-scene-guided cloud-guided; do
for b in 1 2 3 4; do
  ms=$(VSPG_WF_MERGED_BLOCKS=$b timeout -k 10 200 python bench.py --workload $wl --steps 8 --warmup 2 --train-waves 4 --no-cpu-baseline --no-generic --no-pmc --no-fast-arith 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%.3f' % d['ms_per_step'])")
  echo "$wl merged_blocks $b $ms" | tee -a gpurun_out/r05/merged_blocks.txt
done; done
