#!/bin/bash
OUT=$PWD/gpurun_out/pmc_abl
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
i=0
for cfg in "" "usenee=0" "vspguiding=0" "usenee=0 vspguiding=0" "maxdepth=0" "maxdepth=1"; do
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM --output-format csv -d $OUT/c$i -- python3 $REPO/scripts/ablate_one.py $cfg > $OUT/c$i.log 2>&1 || { echo fail $cfg; tail -3 $OUT/c$i.log; }
  python3 - "$cfg" $OUT/c$i <<'PY'
import csv, glob, collections, sys
agg=collections.defaultdict(list)
for f in glob.glob(sys.argv[2]+'/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'k_render_wave' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
d={k:sum(v)/len(v) for k,v in agg.items()}
print("%-24s VALU %.3g SALU %.3g SMEM %.3g wave_cyc %.3g wait %.2f active %.2f laneutil %.2f" % (sys.argv[1] or "default", d['SQ_INSTS_VALU'], d['SQ_INSTS_SALU'], d['SQ_INSTS_SMEM'], d['SQ_WAVE_CYCLES'], d['SQ_WAIT_ANY']/d['SQ_WAVE_CYCLES'], d['SQ_ACTIVE_INST_ANY']/d['SQ_WAVE_CYCLES'], d['SQ_THREAD_CYCLES_VALU']/d['SQ_ACTIVE_INST_VALU']/64))
PY
done
