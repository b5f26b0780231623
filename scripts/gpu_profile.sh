#!/bin/bash
# rocprofv3 kernel-trace stats of the bench + PMC passes (own runs, no trace domains mixed in).
set -o pipefail
TAG=${1:-r01}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py --no-pmc --steps 16 --warmup 2 --no-cpu-baseline > $OUT/bench_trace.log 2>&1 || { echo "trace run failed"; tail -5 $OUT/bench_trace.log; exit 1; }
for pmc in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY" "SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  name=$(echo $pmc | tr ' ' '_' | cut -c1-40)
  timeout -k 10 300 rocprofv3 --pmc $pmc --output-format csv -d $OUT/pmc_$name -- python3 $REPO/bench.py --no-pmc --steps 4 --warmup 1 --no-cpu-baseline > $OUT/pmc_$name.log 2>&1 || { echo "pmc $pmc failed"; tail -3 $OUT/pmc_$name.log; }
done
cd $OUT && find . -name "*.csv" | head -40
