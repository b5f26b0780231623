#!/bin/bash
# shape of the headline kernel's workgroups: 512 threads x 512 paths (two per CU) against 1024 x 1024 (one) and 256 x 256 (four)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03zi
mkdir -p $OUT; : > $OUT/ab.txt
cd $REPO
for rep in 1 2; do
for lib in "" $(ls build/variants/*.so 2>/dev/null); do
  env ${lib:+VSPG_LIB=$PWD/$lib} timeout -k 10 300 python bench.py --steps 32 --warmup 4 --no-cpu-baseline --no-pmc --no-reference-defaults 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('rep$rep ${lib:-default} ms %.4f value %.0f generic %.0f' % (d['ms_per_step'], d['value'], d['generic_instantiation']['value']))" | tee -a $OUT/ab.txt
done; done
