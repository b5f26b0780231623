#!/bin/bash
# A/B: workgroup kernel with the path pool in global memory (POOLG) vs in LDS; unguided headline and guided (VSPG_KERNEL=wg)
mkdir -p gpurun_out; : > gpurun_out/poolg.txt
run() { # name lib args env
  ms=$(env ${2:+VSPG_LIB=$PWD/$2} ${4:+VSPG_KERNEL=$4} timeout -k 10 200 python bench.py --steps 32 --warmup 4 --no-cpu-baseline --no-generic $3 2>gpurun_out/poolg_err.txt | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%.4f ms_step %.4f %s' % (d['roofline']['kernel_ms'], d['ms_per_step'], d['roofline']['kernel']))")
  echo "$1 $ms" | tee -a gpurun_out/poolg.txt
}
for rep in 1 2; do
  run u_default "" "" ""
  for n in u_pg1024 u_pg2048 u_pg4096; do run $n build/variants/lib_$n.so "" ""; done
  if [ -n "$WITH_GUIDED" ]; then
  run g_lane "" "--workload fog-guided" ""
  run g_wg_lds320 "" "--workload fog-guided" wg
  for n in g_pg1024 g_pg2048; do run $n build/variants/lib_$n.so "--workload fog-guided" wg; done
  fi
done
