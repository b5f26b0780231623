#!/bin/bash
# wg2's claim unit: a tile (default), half a tile, a quarter -- correctness of each build, then the headline and guided workloads
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03zq
mkdir -p $OUT; : > $OUT/ab.txt
cd $REPO
for lib in "" build/variants/u5.so build/variants/u4.so; do
  env ${lib:+VSPG_LIB=$PWD/$lib} timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "scheduler or parked or full_size_wave or full_size_guided or training_samples" > $OUT/tests_$(basename ${lib:-default} .so).log 2>&1
  echo "tests ${lib:-default} rc=$?"; tail -2 $OUT/tests_$(basename ${lib:-default} .so).log
done
grep -l "failed\|error" $OUT/tests_*.log && exit 1
for rep in 1 2; do
for lib in "" build/variants/u5.so build/variants/u4.so; do
  env ${lib:+VSPG_LIB=$PWD/$lib} timeout -k 10 300 python bench.py --steps 32 --warmup 4 --no-cpu-baseline --no-pmc 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); rd=d['reference_defaults']; print('rep$rep ${lib:-default} ms %.4f value %.0f generic %.0f | guided trained %.4f training %.3f' % (d['ms_per_step'], d['value'], d['generic_instantiation']['value'], rd['ms_per_trained_wave'], rd['training']['ms_per_wave']))" | tee -a $OUT/ab.txt
done; done
