#!/bin/bash
# round-3 first look: corrected issue-rate microbenchmark, GPU suite sanity, PMC of the guided kernels (lane, wg)
mkdir -p gpurun_out/r03a
O=gpurun_out/r03a
timeout -k 10 300 scripts/microbench/issue > $O/microbench_issue.txt 2>&1; echo "microbench rc=$?"; tail -4 $O/microbench_issue.txt
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $O/tests_gpu.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $O/tests_gpu.log
[ $rc -eq 124 ] && exit 124
for k in lane wg; do
  VSPG_KERNEL=$k BENCH_ARGS="--workload fog-guided --train-waves 8" bash scripts/gpu_pmc3.sh r03_guided_$k > $O/pmc_guided_$k.log 2>&1
  cp gpurun_out/pmc3_r03_guided_$k/summary.txt $O/pmc_guided_$k.txt
done
VSPG_KERNEL=wg bash scripts/gpu_pmc3.sh r03_fog > $O/pmc_fog.log 2>&1; cp gpurun_out/pmc3_r03_fog/summary.txt $O/pmc_fog.txt
timeout -k 10 200 python scripts/prof_sections.py > $O/prof_sections_fog.txt 2>&1
bash scripts/gpu_guided_pmc.sh > $O/guided_ms.txt 2>&1
cat $O/guided_ms.txt
exit 0
