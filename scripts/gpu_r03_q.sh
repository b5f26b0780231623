#!/bin/bash
# HBM traffic of the training wave's kernels (own PMC passes, no trace domains)
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r03q
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export VSPG_TT_TRAIN=10
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/p1 -- python3 $REPO/scripts/train_timing.py > $OUT/p1.log 2>&1 || { tail -5 $OUT/p1.log; exit 1; }
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/p2 -- python3 $REPO/scripts/train_timing.py > $OUT/p2.log 2>&1 || { tail -5 $OUT/p2.log; exit 1; }
python3 $REPO/scripts/summarize_pmc.py $OUT > $OUT/summary.txt
python3 - <<PY
import csv, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(anonymous namespace\)::|vspg::", "", re.sub(r"^void ", "", r["Kernel_Name"]))
        k = k.split("(")[0]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(agg.items()):
    f = cs.get("FETCH_SIZE", [0]); w = cs.get("WRITE_SIZE", [0])
    print("%-90s n=%3d fetch(x2) %8.1f MB  write %8.1f MB" % (k[:90], len(f), 2 * sum(f) / len(f) * 1024 / 1e6, sum(w) / len(w) * 1024 / 1e6))
PY
