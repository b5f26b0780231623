#!/bin/bash
# wave-state / lane-utilisation counters for the render kernel (VSPG_KERNEL selects which)
set -o pipefail
TAG=${1:-x}
OUT=$PWD/gpurun_out/pmc3_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
i=0
for pmc in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $pmc --output-format csv -d $OUT/p$i -- python3 $REPO/bench.py --no-pmc --steps 3 --warmup ${BENCH_WARMUP:-1} --no-cpu-baseline $BENCH_ARGS > $OUT/p$i.log 2>&1 || { echo "pmc $pmc failed"; tail -3 $OUT/p$i.log; }
done
python3 - <<PY | tee $OUT/summary.txt
import csv, glob, collections
for f in sorted(glob.glob('$OUT/p*/*/*_counter_collection.csv')):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'k_render_wave' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items(): print(k, "%.5g"%(sum(v)/len(v)))
PY
