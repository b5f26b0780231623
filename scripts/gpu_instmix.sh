#!/bin/bash
# Dynamic instruction mix of a bench workload's kernels (verdict r4 item 1a): rocprofv3 --pmc passes, own runs, never mixed with
# trace domains; per-kernel table by scripts/summarize_instmix.py -> profiles/<tag>_instmix_<workload>.txt
#   bash scripts/gpu_instmix.sh <tag> <workload> [extra bench args]
set -o pipefail
TAG=${1:-r05}; WL=${2:-fog}; shift 2
OUT=$PWD/gpurun_out/instmix_${TAG}_$WL
rm -rf $OUT; mkdir -p $OUT
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
i=0
for pmc in \
 "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SALU SQ_INSTS_LDS" \
 "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_BRANCH" \
 "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
 "SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $pmc --output-format csv -d $OUT/p$i -- python3 $REPO/bench.py --workload $WL --steps 2 --warmup 1 --train-waves 6 --no-cpu-baseline --no-generic --no-pmc --no-reference-defaults "$@" > $OUT/p$i.log 2>&1 || { echo "pmc pass $i failed"; tail -3 $OUT/p$i.log; }
done
python3 $REPO/scripts/summarize_instmix.py $OUT 3 > $REPO/gpurun_out/${TAG}_instmix_${WL}.txt; cat $REPO/gpurun_out/${TAG}_instmix_${WL}.txt
