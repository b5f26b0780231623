#!/bin/bash
# per-kernel time of every training update (rocprofv3 kernel trace of scripts/train_timing.py)
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/trainprof
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $REPO/scripts/train_timing.py > $OUT.log 2>&1 || { tail -5 $OUT.log; exit 1; }
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/*/*_kernel_trace.csv")[0]
ups = []; cur = None
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if "k_train" not in k: continue
    name = k[k.index("k_train_") + 8:].split("(")[0][:12]
    if name.startswith("decay"): cur = collections.OrderedDict(); ups.append(cur)   # first kernel of an update
    if cur is None: continue
    cur[name] = cur.get(name, 0) + (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
ren = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if "k_render_wave" in k or "k_propagate" in k or "k_field_aux" in k:
        name = ("render<train>" if ", true>(" in k else "render<guided>") if "k_render_wave" in k else k.split("::")[-1].split("(")[0]
        ren[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for k, v in ren.items(): print("%-16s n=%d mean %.3f ms" % (k, len(v), sum(v) / len(v)))
for i, u in enumerate(ups):
    print(i, "total %.2f ms: " % sum(u.values()), " ".join("%s=%.2f" % kv for kv in u.items()))
PY
grep "^wave" $OUT.log
