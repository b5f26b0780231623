"""Dynamic instruction mix per kernel from the rocprofv3 --pmc passes of scripts/gpu_instmix.sh:
    python scripts/summarize_instmix.py <dir with p*/.../*counter_collection.csv> [waves profiled]
Counter values are summed over a kernel's dispatches and divided by the number of waves (vspg_render_wave calls) profiled, i.e.
wave-instructions per 1080p wave.  A float division here is v_div_scale x2 + v_rcp_f32 + 4 FMA + v_div_fmas + v_div_fixup (~10.5
vector instructions, scripts/static_instmix.py); every one holds exactly one transcendental (v_rcp_f32), so TRANS_F32 bounds the
number of divisions from above (sqrt, FastExp's exp2-free polynomial has none, v_rsq / v_sqrt count too)."""
import collections, csv, glob, os, re, sys
src = sys.argv[1]
waves = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
tot = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(os.path.join(src, "p*", "**", "*counter_collection.csv"), recursive=True):
    rows = list(csv.DictReader(open(f)))
    last_train = max([int(r["Dispatch_Id"]) for r in rows if "k_propagate" in r["Kernel_Name"] or "k_train_" in r["Kernel_Name"]], default=-1)
    for r in rows:
        if int(r["Dispatch_Id"]) <= last_train and ("k_render_wave" in r["Kernel_Name"] or "k_wf_" in r["Kernel_Name"]):
            continue
        k = re.sub(r"\(anonymous namespace\)::|vspg::(?:fastw::|fast::)?", "", re.sub(r"^void ", "", r["Kernel_Name"])).split("(")[0]
        if "::fastw::" in r["Kernel_Name"] or "::fast::" in r["Kernel_Name"]:  # the tolerance-mode instantiations (csrc/vspg_arith.h) under their own names
            k = k.split("<")[0] + (".fastw" if "::fastw::" in r["Kernel_Name"] else ".fast") + k[len(k.split("<")[0]):]
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        launches[k][r["Counter_Name"]] += 1
G = lambda m, c: m.get(c, 0.0) / waves
print("wave-instructions per 1080p wave (sum over the kernel's launches / %g waves); share = of SQ_INSTS_VALU" % waves)
for k, m in sorted(tot.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0)):
    if not k.startswith("k_") or m.get("SQ_INSTS_VALU", 0) < 1e5:
        continue
    valu = G(m, "SQ_INSTS_VALU")
    n = max(launches[k].values()) / waves
    print("\n%s   (%.1f launches per wave)" % (k[:110], n))
    rows = [("VALU total", valu)]
    for c in ("ADD_F32", "MUL_F32", "FMA_F32", "TRANS_F32", "ADD_F64", "MUL_F64", "FMA_F64", "TRANS_F64", "INT32", "INT64", "CVT"):
        rows.append(("  " + c, G(m, "SQ_INSTS_VALU_" + c)))
    known = sum(v for _, v in rows[1:])
    rows.append(("  other (mov / select / compare / bit ops / lane ops)", valu - known))
    f64 = sum(G(m, "SQ_INSTS_VALU_%s_F64" % c) for c in ("ADD", "MUL", "FMA", "TRANS"))
    for name, v in rows:
        print("  %-56s %12.4g  %6.3f" % (name, v, v / valu if valu else 0))
    tr = G(m, "SQ_INSTS_VALU_TRANS_F32")
    print("  %-56s %12.4g  %6.3f   (<= TRANS_F32 divisions x 10.5 instructions)" % ("division expansions, upper bound", tr * 10.5, tr * 10.5 / valu if valu else 0))
    print("  %-56s %12.4g  %6.3f   (half rate: counted twice in issue slots)" % ("FP64 total", f64, f64 / valu if valu else 0))
    for c in ("SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_BRANCH"):
        print("  %-56s %12.4g  %6.3f" % (c, G(m, c), G(m, c) / valu if valu else 0))
    wc = G(m, "SQ_WAVE_CYCLES")
    if wc:
        for c in ("SQ_WAIT_INST_LDS", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_INST_CYCLES_VMEM_RD",
                  "SQ_INST_CYCLES_VMEM_WR", "SQ_INST_CYCLES_SALU", "SQ_LDS_BANK_CONFLICT"):
            print("  %-56s %12.4g  %6.3f of SQ_WAVE_CYCLES" % (c, G(m, c), G(m, c) / wc))
    if m.get("SQ_ACTIVE_INST_VALU"):
        print("  %-56s %12.3f" % ("lane utilisation (THREAD_CYCLES_VALU / 64 ACTIVE_INST_VALU)", m.get("SQ_THREAD_CYCLES_VALU", 0) / (64 * m["SQ_ACTIVE_INST_VALU"])))
