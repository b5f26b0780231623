#!/bin/bash
# A/B of library builds (build/variants/*.so vs the default) on given workloads: ms per step, same box
mkdir -p gpurun_out; : > gpurun_out/variants_wl.txt
for rep in 1 2; do
for wl in ${WORKLOADS:-cloud cloud-nvdb}; do
for lib in "" $(ls build/variants/*.so 2>/dev/null); do
  ms=$(env ${lib:+VSPG_LIB=$PWD/$lib} timeout -k 10 200 python bench.py --workload $wl --steps 8 --warmup 2 --train-waves 4 --no-cpu-baseline --no-generic 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%.3f' % d['ms_per_step'])")
  echo "rep$rep $wl ${lib:-default} $ms" | tee -a gpurun_out/variants_wl.txt
done; done; done
