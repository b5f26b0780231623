#!/bin/bash
# A/B of library builds on the headline workload (kernel ms from the bench's HIP events), same box
mkdir -p gpurun_out; : > gpurun_out/variants.txt
for rep in 1 2; do
for lib in "" $(ls build/variants/*.so 2>/dev/null); do
  ms=$(VSPG_LIB=${lib:+$PWD/$lib} timeout -k 10 120 python bench.py --steps 48 --warmup 4 --no-cpu-baseline --no-generic ${VARIANT_ARGS} 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%.4f %s' % (d['roofline']['kernel_ms'], d['roofline']['kernel']))")
  echo "rep$rep ${lib:-default} $ms" | tee -a gpurun_out/variants.txt
done; done
