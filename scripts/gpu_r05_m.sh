#!/bin/bash
# round 5, step m: advance rounds per tracking step (VSPG_WF_ROUNDS) and refill threshold (VSPG_WF_REFILL) on the boundary scene,
# whose walks cross many majorant cells per tentative collision (8 density queries per path against the dense cloud's 49)
mkdir -p gpurun_out/r05; : > gpurun_out/r05/rounds_sweep.txt
for wl in cloud-scene cloud-scene-nvdb cloud; do
for r in 2 3 4 6 8 12; do
  ms=$(VSPG_WF_ROUNDS=$r timeout -k 10 200 python bench.py --workload $wl --steps 8 --warmup 2 --no-cpu-baseline --no-generic --no-pmc --no-fast-arith 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%.3f' % d['ms_per_step'])")
  echo "$wl rounds $r $ms" | tee -a gpurun_out/r05/rounds_sweep.txt
done
for f in 8 24 32; do
  ms=$(VSPG_WF_REFILL=$f timeout -k 10 200 python bench.py --workload $wl --steps 8 --warmup 2 --no-cpu-baseline --no-generic --no-pmc --no-fast-arith 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%.3f' % d['ms_per_step'])")
  echo "$wl refill $f $ms" | tee -a gpurun_out/r05/rounds_sweep.txt
done
done
