#!/bin/bash
# kd staging sizes for the guided workgroup kernel, then its PMC passes
mkdir -p gpurun_out/r03h
O=gpurun_out/r03h
for lib in "" build/variants/kd128.so build/variants/kd0.so; do
  for ng in 0 1; do
  export VSPG_LIB=${lib:+$PWD/$lib}; [ -z "$lib" ] && unset VSPG_LIB
  if [ $ng = 1 ]; then export VSPG_NO_GREY_GUIDED=1; else unset VSPG_NO_GREY_GUIDED; fi
  timeout -k 10 300 python bench.py --workload fog-guided --steps 16 --warmup 2 --train-waves 8 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('${lib:-default} nogrey=$ng', d['roofline']['kernel'], 'kernel_ms %.3f' % d['roofline']['kernel_ms'], 'train ms %.2f' % d['training']['ms_per_wave'])" | tee -a $O/ab.txt
  done
done
unset VSPG_LIB VSPG_NO_GREY_GUIDED
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "guided_workgroup" > $O/t.log 2>&1; echo "parity rc=$?"; tail -2 $O/t.log
BENCH_ARGS="--workload fog-guided --train-waves 8" BENCH_WARMUP=2 bash scripts/gpu_pmc3.sh r03_fogguided > $O/pmc.log 2>&1
python3 scripts/summarize_pmc.py gpurun_out/pmc3_r03_fogguided | head -6 | tee $O/pmc_fog-guided.txt
exit 0
