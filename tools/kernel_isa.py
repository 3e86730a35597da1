#!/usr/bin/env python3
"""ISA of kernels of meao_kernels.hip, compiled like the product (no GPU needed).

    python tools/kernel_isa.py <regex on the demangled name> [-DFLAG ...] [--stats]

Prints each matching kernel's instructions; --stats prints instruction-class counts instead (static counts of the
whole kernel body: VALU by class, SALU, LDS, VMEM, waitcnt), which is how the integer / address arithmetic of a
kernel is found without a PMC pass."""
import collections, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from miniengineao_amd import build

_CACHE = os.path.join(tempfile.gettempdir(), "meao_kernel_isa")


def asm_text(flags):
    src = os.path.join(ROOT, "miniengineao_amd", "csrc", "meao_kernels.hip")
    csrc = os.path.dirname(src)       # the unity file includes every kernel unit and device header: any of them invalidates the cache
    key = str(abs(hash((max(os.path.getmtime(os.path.join(csrc, f)) for f in os.listdir(csrc)), tuple(flags)))))
    out = os.path.join(_CACHE, key + ".s")
    if not os.path.exists(out):
        os.makedirs(_CACHE, exist_ok=True)
        base = [f for f in build.FLAGS if f not in ("-shared", "-fPIC", "-fvisibility=hidden")]
        subprocess.run([build.hipcc(), *base, *flags, "-I" + os.path.join(ROOT, "include"), "--cuda-device-only", "-S", src, "-o", out],
                       check=True, capture_output=True)
    return open(out).read()


def kernels(text):
    names = re.findall(r"^(_ZN4meao\w+):", text, re.M)
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    out = {}
    for mangled, nice in zip(names, dem):
        m = re.search(r"^%s:[^\n]*\n(.*?)^\.Lfunc_end" % re.escape(mangled), text, re.S | re.M)
        if m:
            body = [l.split(";")[0].strip() for l in m.group(1).splitlines()]
            out[nice] = [l for l in body if l and not l.startswith(".") and not l.endswith(":")]
    return out


def classify(op):
    if op.startswith("v_"):
        if re.match(r"v_(rcp|rsq|sqrt|exp|log|sin|cos)_", op): return "valu_trans"
        if re.match(r"v_(fma|fmac|mad|mac)_f32|v_fma_mix", op): return "valu_fma_f32"
        if re.match(r"v_(add|sub|subrev)_f32", op): return "valu_add_f32"
        if re.match(r"v_mul_f32", op): return "valu_mul_f32"
        if re.match(r"v_pk_", op): return "valu_pk"
        if re.match(r"v_cvt_", op): return "valu_cvt"
        if re.match(r"v_(max|min|med3|max3|min3|fract|cndmask|cmp|cmpx)", op) and not re.search(r"_[iu](16|32|64)", op): return "valu_f32_other"
        if re.search(r"_[iu]64|_u64|lshl_add_u64|mad_[iu]64", op): return "valu_int64"
        if re.search(r"_[iu](16|32)|_b32|_u24|_i24|lshl|lshr|ashr|and|or|xor|bfe|bfi|perm|add_co|addc|sub_co|subb|add3|mad_u|mul_lo|mul_hi|alignbit|sad", op): return "valu_int32"
        if re.match(r"v_mov|v_readlane|v_readfirstlane|v_writelane|v_swap|v_accvgpr|v_nop", op): return "valu_mov"
        return "valu_other"
    if op.startswith("s_waitcnt"): return "waitcnt"
    if op.startswith("s_barrier"): return "barrier"
    if op.startswith("s_load") or op.startswith("s_buffer"): return "smem"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_") or op.startswith("scratch_"): return "vmem"
    return "other"


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("-")]
    flags = [a for a in sys.argv[1:] if a.startswith("-D")]
    stats = "--stats" in sys.argv
    pat = re.compile(args[0] if args else ".")
    for name, body in kernels(asm_text(flags)).items():
        if not pat.search(name):
            continue
        if stats:
            c = collections.Counter(classify(l.split()[0]) for l in body)
            valu = sum(v for k, v in c.items() if k.startswith("valu"))
            print(f"{name}: {len(body)} instructions, VALU {valu} " + " ".join(f"{k}={v}" for k, v in sorted(c.items())))
        else:
            print(f"=== {name} ({len(body)} instructions)")
            print("\n".join(body))


if __name__ == "__main__":
    main()
