#!/bin/bash
set -o pipefail
TAG=${1:-x}
OUT=$PWD/gpurun_out/pmc2_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
i=0
for pmc in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_INSTS_BRANCH" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F64" "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $pmc --output-format csv -d $OUT/p$i -- python3 $REPO/bench.py --no-pmc --steps 3 --warmup 1 --no-cpu-baseline $BENCH_ARGS > $OUT/p$i.log 2>&1 || { echo "pmc $pmc failed"; tail -3 $OUT/p$i.log; }
done
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob('$OUT/p*/*/*_counter_collection.csv')):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'k_render_wave' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items(): print(k, "%.4g"%(sum(v)/len(v)))
PY
