#!/bin/bash
mkdir -p gpurun_out/r03n
O=$PWD/gpurun_out/r03n
REPO=${GRAFT_REPO_ROOT:-/root/repo}
timeout -k 10 120 $REPO/scripts/microbench/gather32 > $O/gather32_timing.txt 2>&1; cat $O/gather32_timing.txt
cd /tmp && export TMPDIR=/tmp
for pmc in "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf $O/pmc_$pmc; timeout -k 10 200 rocprofv3 --pmc $pmc --output-format csv -d $O/pmc_$pmc -- $REPO/scripts/microbench/gather32 > $O/pmc_$pmc.log 2>&1 || echo "pmc $pmc failed"
done
python3 - <<PY | tee $O/gather32_pmc.txt
import csv, glob, collections
for c in ("FETCH_SIZE","WRITE_SIZE"):
    for f in glob.glob('$O/pmc_%s/**/*counter_collection.csv' % c, recursive=True):
        agg=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            agg[(r['Kernel_Name'].split('(')[0], r['Counter_Name'])].append(float(r['Counter_Value']))
        for k,v in sorted(agg.items()): print(k[0], k[1], 'KiB per launch: %.0f' % (sum(v)/len(v)), ' -> bytes per record (2^26 records): %.1f' % (sum(v)/len(v)*1024/2**26))
PY
exit 0
