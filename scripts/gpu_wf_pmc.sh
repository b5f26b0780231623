#!/bin/bash
# PMC passes (own runs, no trace domains mixed in) of a bench workload; per-kernel means by scripts/summarize_pmc.py
set -o pipefail
TAG=${1:-wf}; WL=${2:-cloud}
OUT=$PWD/gpurun_out/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
i=0
for pmc in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY" "SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $pmc --output-format csv -d $OUT/p$i -- python3 $REPO/bench.py --workload $WL --steps 2 --warmup 1 --train-waves 6 --no-cpu-baseline --no-generic --no-pmc --no-reference-defaults > $OUT/p$i.log 2>&1 || { echo "pmc pass $i failed"; tail -3 $OUT/p$i.log; }
done
python3 $REPO/scripts/summarize_pmc.py $OUT > $OUT/summary.txt; cat $OUT/summary.txt
