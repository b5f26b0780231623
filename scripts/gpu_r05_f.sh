#!/bin/bash
# round 5: timeline of a training wave (kernel trace with timestamps): where the time between the kernels goes
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r05/trainprof
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
VSPG_TT_TRAIN=12 VSPG_TRAIN_GRAPH=${VSPG_TRAIN_GRAPH:-0} timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT -- python3 $REPO/scripts/train_timing.py > $OUT.log 2>&1 || { tail -5 $OUT.log; exit 1; }
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/*/*_kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# the last training wave: from the last 'train' render kernel to the next render kernel
idx = [i for i, r in enumerate(rows) if "k_render_wave" in r["Kernel_Name"]]
tr = [i for i in idx if ", true>(" in rows[i]["Kernel_Name"]]
a = tr[-2]; b = tr[-1]
t0 = int(rows[a]["Start_Timestamp"])
prev_end = t0
for r in rows[a:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:40]
    print("%-42s start %8.1f us  dur %7.1f us  gap before %6.1f us" % (k, (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3))
    prev_end = e
PY
grep "^wave" $OUT.log | tail -8
