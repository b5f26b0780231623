"""Static instruction mix of the gfx950 code objects in an object file / shared library.

    python scripts/static_instmix.py vspg-pbrt-v4_amd/csrc/vspg_capi.o [substring of the demangled kernel name ...]

Per kernel: vector / scalar / LDS / global instruction counts and, inside the vector ones, the classes the dynamic counters of
scripts/gpu_instmix.sh report (F32 add / mul / fma, F64, integer, transcendental, conversions, moves / selects / compares) plus the
number of IEEE-754 float divisions (one `v_div_fmas_f32` each: v_div_scale x2 + v_rcp + 4 FMA + v_div_fmas + v_div_fixup ~ 10.5
vector instructions) and double divisions (`v_div_fmas_f64`).  Static counts say what the compiler emitted, not what runs -- the
dynamic mix is profiles/r05_instmix_*.txt."""
import collections
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def code_objects(path):
    tmp = tempfile.mkdtemp(prefix="instmix_")
    local = os.path.join(tmp, os.path.basename(path))
    os.symlink(os.path.abspath(path), local)
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return [os.path.join(tmp, f) for f in sorted(os.listdir(tmp)) if "amdgcn" in f]


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return dict(zip(names, out))


CLASSES = [
    ("div_f32", re.compile(r"^v_div_fmas_f32")),
    ("div_f64", re.compile(r"^v_div_fmas_f64")),
    ("trans_f32", re.compile(r"^v_(rcp|rsq|sqrt|exp|log|sin|cos)_(f32|iflag_f32|legacy_f32)")),
    ("trans_f64", re.compile(r"^v_(rcp|rsq|sqrt)_f64")),
    ("f64", re.compile(r"^v_\w+_f64")),
    ("fma_f32", re.compile(r"^v_(fma|fmac|mad|mac|pk_fma|fmaak|fmamk|div_fmas|div_fixup)_f32")),
    ("mul_f32", re.compile(r"^v_(mul|pk_mul|mul_legacy|ldexp)_f32")),
    ("add_f32", re.compile(r"^v_(add|sub|subrev|pk_add)_f32")),
    ("minmax_f32", re.compile(r"^v_(min|max|med3|min3|max3)_f32")),
    ("cvt", re.compile(r"^v_(cvt|frexp|fract|floor|ceil|trunc|rndne|div_scale)_")),
    ("cmp", re.compile(r"^v_cmp")),
    ("mov_sel", re.compile(r"^v_(mov|cndmask|readlane|readfirstlane|writelane|swap|accvgpr|perm|bfi|bfe|alignbit|mbcnt)")),
    ("int", re.compile(r"^v_(add|sub|subrev|mul|mad|lshl|lshr|ashr|and|or|xor|not|add3|lshl_add|lshl_or|and_or|or3|xad|min|max|addc|subb|mul_hi|mul_lo|lshlrev|lshrrev|ashrrev|bcnt|ffbh|ffbl|sad|xnor|add_lshl|xor3|subbrev)_?(co_)?(u|i|b)?(16|24|32|64)?")),
]


def classify(op):
    for name, rx in CLASSES:
        if rx.match(op):
            return name
    return "other_v"


def main():
    path = sys.argv[1]
    filters = sys.argv[2:]
    for co in code_objects(path):
        text = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", co], capture_output=True, text=True).stdout
        funcs = collections.OrderedDict()
        cur = None
        for line in text.split("\n"):
            m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
            if m:
                cur = m.group(1)
                funcs[cur] = collections.Counter()
                continue
            m = re.match(r"^\s+([a-z_0-9]+)", line)
            if cur is None or not m:
                continue
            op = m.group(1)
            c = funcs[cur]
            if op.startswith("v_"):
                c["valu"] += 1
                c[classify(op)] += 1
            elif op.startswith("s_"):
                c["salu"] += 1
                if op.startswith("s_waitcnt"):
                    c["waitcnt"] += 1
                if op.startswith("s_barrier"):
                    c["barrier"] += 1
                if op.startswith("s_load") or op.startswith("s_buffer_load"):
                    c["smem"] += 1
            elif op.startswith("ds_"):
                c["lds"] += 1
            elif op.startswith(("global_", "flat_", "buffer_")):
                c["vmem"] += 1
            elif op.startswith("scratch_"):
                c["scratch"] += 1
        names = demangle(list(funcs))
        cols = ["valu", "div_f32", "div_f64", "trans_f32", "f64", "fma_f32", "mul_f32", "add_f32", "minmax_f32", "cvt", "cmp", "mov_sel", "int", "other_v",
                "salu", "lds", "vmem", "scratch", "barrier"]
        print("%-72s " % "kernel" + " ".join("%9s" % c for c in cols) + "   div share")
        for f, c in funcs.items():
            dn = re.sub(r"\(anonymous namespace\)::|vspg::", "", names.get(f, f))
            dn = re.sub(r"^void ", "", dn).split("(")[0]
            if filters and not any(s in dn for s in filters):
                continue
            if c["valu"] < 50:
                continue
            share = (10.5 * c["div_f32"]) / max(1, c["valu"])
            print("%-72s " % dn[:72] + " ".join("%9d" % c[k] for k in cols) + "   %.3f" % share)


if __name__ == "__main__":
    main()
