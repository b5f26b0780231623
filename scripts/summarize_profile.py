"""Copy the judged summaries of one scripts/gpu_profile.sh run into profiles/.

  python scripts/summarize_profile.py gpurun_out/prof_<tag> <tag>

writes profiles/<tag>_kernel_stats_bench_steps16.csv (rocprofv3 --kernel-trace --stats, verbatim) and
profiles/<tag>_pmc_k_render_wave.json (per-launch means of every PMC counter for the render kernel;
FETCH_SIZE / WRITE_SIZE come from their own passes, KiB per launch as rocprofv3 reports them).
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

src, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_dir = os.path.join(root, "profiles")
os.makedirs(out_dir, exist_ok=True)
stats = glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))
if stats:
    shutil.copy(stats[0], os.path.join(out_dir, "%s_kernel_stats_bench_steps16.csv" % tag))
agg = collections.defaultdict(list)
kernel = None
for f in glob.glob(os.path.join(src, "pmc_*", "*", "*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "k_render_wave" in r["Kernel_Name"]:
            kernel = r["Kernel_Name"].split("(")[1] if r["Kernel_Name"].startswith("void (") else r["Kernel_Name"]
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {k: {"launches": len(v), "mean_per_launch": sum(v) / len(v)} for k, v in sorted(agg.items())}
res["_kernel"] = [r["Kernel_Name"] for f in glob.glob(os.path.join(src, "pmc_*", "*", "*_counter_collection.csv"))
                  for r in csv.DictReader(open(f)) if "k_render_wave" in r["Kernel_Name"]][:1]
json.dump(res, open(os.path.join(out_dir, "%s_pmc_k_render_wave.json" % tag), "w"), indent=1)
print(json.dumps({k: v["mean_per_launch"] for k, v in res.items() if not k.startswith("_")}, indent=1))
