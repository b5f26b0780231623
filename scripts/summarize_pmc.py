"""Per-kernel means of rocprofv3 --pmc passes: python scripts/summarize_pmc.py <dir with p*/.../*counter_collection.csv> [out.json]
Derived: lane_util = SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU); wait = SQ_WAIT_ANY / SQ_WAVE_CYCLES;
hbm_bytes = 2 * FETCH_SIZE KiB (gfx950: 128-B reads tallied at 64 B) + WRITE_SIZE KiB (MI355X_MICROARCH.md, HBM)."""
import collections, csv, glob, json, os, re, sys
src = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(src, "p*", "**", "*counter_collection.csv"), recursive=True):
    rows = list(csv.DictReader(open(f)))
    # a guided run trains its field first: the path kernels of the TRAINED waves are the launches behind the last training kernel
    # (k_propagate, k_train_*; dispatch order) -- as bench.py's live PMC counts them.  Training kernels themselves are kept.
    last_train = max([int(r["Dispatch_Id"]) for r in rows if "k_propagate" in r["Kernel_Name"] or "k_train_" in r["Kernel_Name"]], default=-1)
    for r in rows:
        if int(r["Dispatch_Id"]) <= last_train and ("k_render_wave" in r["Kernel_Name"] or "k_wf_" in r["Kernel_Name"]):
            continue
        k = re.sub(r"\(anonymous namespace\)::|vspg::(?:fastw::|fast::)?", "", re.sub(r"^void ", "", r["Kernel_Name"])).split("(")[0]
        if "::fastw::" in r["Kernel_Name"] or "::fast::" in r["Kernel_Name"]:  # the tolerance-mode instantiations (csrc/vspg_arith.h) under their own names
            k = k.split("<")[0] + (".fastw" if "::fastw::" in r["Kernel_Name"] else ".fast") + k[len(k.split("<")[0]):]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, cs in agg.items():
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    m["launches"] = max(len(v) for v in cs.values())
    if m.get("SQ_ACTIVE_INST_VALU"):
        m["lane_util"] = m.get("SQ_THREAD_CYCLES_VALU", 0) / (64 * m["SQ_ACTIVE_INST_VALU"])
    if m.get("SQ_WAVE_CYCLES"):
        m["wait_frac"] = m.get("SQ_WAIT_ANY", 0) / m["SQ_WAVE_CYCLES"]
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        m["hbm_bytes"] = (2 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024
    res[k] = m
if len(sys.argv) > 2:
    # profiles/<tag>_pmc_<workload>.json as bench.py reads it: per-wave totals over the workload's kernels + the hash of the
    # kernel sources the counters were taken on (bench.py reports `traffic` only while the sources are unchanged)
    import hashlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = os.path.join(root, "vspg-pbrt-v4_amd", "csrc")
    h = hashlib.sha256()
    for n in sorted(os.listdir(d)):
        if n.endswith((".h", ".hip")) or n == "Makefile":
            h.update(n.encode())
            h.update(open(os.path.join(d, n), "rb").read())
    workload = sys.argv[3] if len(sys.argv) > 3 else "fog"
    waves = int(sys.argv[4]) if len(sys.argv) > 4 else 3
    path_kernels = {k: m for k, m in res.items() if k.startswith(("k_wf_", "k_render_wave"))}
    out = {"_csrc_hash": h.hexdigest()[:16], "_workload": workload, "_res": [1920, 1080], "_waves_profiled": waves, "kernels": res}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        out[c] = {"mean_per_launch": sum(m.get(c, 0) * m["launches"] for m in path_kernels.values()) / waves,
                  "note": "KiB per WAVE (one vspg_render_wave call), summed over its path kernels"}
    json.dump(out, open(sys.argv[2], "w"), indent=1, sort_keys=True)
for k, m in sorted(res.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", 0)):
    if not k.startswith("k_"):
        continue
    print("%-46s n=%3d valu %.3e lane_util %.3f wait %.3f hbm %.1f MB  valu/wave %.0f" % (
        k[:46], m["launches"], m.get("SQ_INSTS_VALU", 0), m.get("lane_util", 0), m.get("wait_frac", 0), m.get("hbm_bytes", 0) / 1e6,
        m.get("SQ_INSTS_VALU", 0) / max(1, m.get("SQ_WAVES", 1))))
