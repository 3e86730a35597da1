#!/usr/bin/env python3
"""Instruction-level diff of two builds of meao_kernels.hip (no GPU needed): proves that a refactoring of the
kernel source is codegen-neutral, or lists the kernels it touched.

    python tools/isa_diff.py <git-rev | path/to/meao_kernels.hip> [-DFLAG ...]     # against the working tree
"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from miniengineao_amd import build

def asm_of(src_dir, flags, units=("meao_kernels.hip",)):
    """units: the translation units to compile; the working tree is compiled the way the product is (build.KERNEL_UNITS, in
    parallel), a reference revision as whatever it has (the single meao_kernels.hip before round 5)."""
    from concurrent.futures import ThreadPoolExecutor
    base = [f for f in build.FLAGS if f not in ("-shared", "-fPIC", "-fvisibility=hidden")]
    with tempfile.TemporaryDirectory() as tmp:
        def one(u):
            out = os.path.join(tmp, u + ".s")
            subprocess.run([build.hipcc(), *base, *flags, "-I" + os.path.join(ROOT, "include"), "-I" + src_dir, "--cuda-device-only", "-S",
                            os.path.join(src_dir, u), "-o", out], check=True, capture_output=True)
            return open(out).read()
        with ThreadPoolExecutor(os.cpu_count() or 1) as ex:
            text = "\n".join(ex.map(one, units))
    funcs = {}
    for m in re.finditer(r"^(_ZN4meao\w+):[^\n]*\n(.*?)^\.Lfunc_end", text, re.S | re.M):
        body = [l.split(";")[0].strip() for l in m.group(2).splitlines()]
        funcs[m.group(1)] = [re.sub(r"\.LBB\d+_\d+", "L", l) for l in body if l and not l.startswith(".")]
    return funcs

def main():
    ref, flags = sys.argv[1], sys.argv[2:]
    csrc = os.path.join(ROOT, "miniengineao_amd", "csrc")
    with tempfile.TemporaryDirectory() as tmp:
        if os.path.exists(ref):
            old_dir = os.path.dirname(os.path.abspath(ref))
        else:
            old_dir = tmp
            listing = subprocess.run(["git", "ls-tree", "--name-only", f"{ref}:miniengineao_amd/csrc/"], cwd=ROOT, check=True,
                                     capture_output=True, text=True).stdout.split()
            for f in listing:
                if f.endswith((".hip", ".hpp")):
                    open(os.path.join(tmp, f), "w").write(subprocess.run(
                        ["git", "show", f"{ref}:miniengineao_amd/csrc/{f}"], cwd=ROOT, check=True, capture_output=True, text=True).stdout)
            src = open(os.path.join(tmp, "meao_kernels.hpp")).read().replace('"../../include/meao.h"', '"meao.h"')
            open(os.path.join(tmp, "meao_kernels.hpp"), "w").write(src)
        old_units = [u for u in build.KERNEL_UNITS if os.path.exists(os.path.join(old_dir, u))] or ["meao_kernels.hip"]
        a, b = asm_of(old_dir, flags, old_units), asm_of(csrc, flags, build.KERNEL_UNITS)
    changed = [k for k in sorted(set(a) & set(b)) if a[k] != b[k]]
    print(f"{len(set(a) & set(b)) - len(changed)} kernels identical, {len(changed)} changed, "
          f"{len(set(a) - set(b))} only in {ref}, {len(set(b) - set(a))} only in the working tree")
    names = subprocess.run(["c++filt"], input="\n".join(changed), capture_output=True, text=True).stdout.splitlines()
    strip = re.compile(r"^void |meao::\(anonymous namespace\)::")
    for k, n in zip(changed, names):
        print("  %6d -> %6d instructions  %s" % (len(a[k]), len(b[k]), strip.sub("", n)[:110]))
    return 0

if __name__ == "__main__":
    sys.exit(main())
