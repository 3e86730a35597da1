#!/usr/bin/env python3
"""Every kernel instantiation of the product library, by the configuration that reaches it.

    python tools/kernel_instantiations.py > profiles/r05_kernel_instantiations.txt

Template arguments <AOFMT, RTNE, DIV, ...>: AOFMT 0 = R8 / 1 = F16 AO storage (meao_config::ao_format), RTNE =
meao_config::f16_rounding, DIV 0 = exact v_rcp_f32 sequences (+ the IEEE body for hostile frames inside the same kernel),
1 = IEEE division only (RTNE storage, or tolerances outside the exact sequences' verified range).
A context reaches exactly one (AOFMT, RTNE, DIV) column; which launch shapes of it run is decided per call by size."""
import collections, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_diff
from miniengineao_amd import build

funcs = isa_diff.asm_of(os.path.join(ROOT, "miniengineao_amd", "csrc"), [], build.KERNEL_UNITS)
names = subprocess.run(["c++filt"], input="\n".join(funcs), capture_output=True, text=True).stdout.splitlines()
strip = re.compile(r"^void |meao::\(anonymous namespace\)::|\(.*$")
COLUMN = {("0", "false", "0"): "R8 RTZ exact   (DEFAULT context)", ("0", "false", "1"): "R8 RTZ IEEE    (tolerances outside the exact range)",
          ("0", "true", "1"): "R8 RTNE IEEE   (f16_rounding = RTNE)",
          ("1", "false", "0"): "F16 RTZ exact  (BASELINE config 5)", ("1", "false", "1"): "F16 RTZ IEEE", ("1", "true", "1"): "F16 RTNE IEEE"}
by_col = collections.defaultdict(list)
other = []
for mangled, n in zip(funcs, names):
    n = strip.sub("", n)
    m = re.match(r"(\w+)<(.*)>$", n)
    args = [a.strip() for a in m.group(2).split(",")] if m else []
    base = m.group(1) if m else n
    key = None
    if base.startswith(("render", "upsample")) and len(args) >= 3:
        key = (args[0], args[1], args[2])
    elif base.startswith("downsample") and len(args) >= 3:          # <VEC, DIV, ROWS>: no f16 store in the pass
        key = ("any", "-", args[1])
    (by_col[key].append((n, len(funcs[mangled]))) if key else other.append((n, len(funcs[mangled]))))
total = sum(len(v) for v in by_col.values()) + len(other)
print(f"{total} kernel instantiations in libmeao_hip.so ({os.path.getsize(build.LIB_PATH) / 1e6:.2f} MB); instructions = static ISA length\n")
for key in sorted(by_col, key=str):
    title = COLUMN.get(key, f"downsample pass, DIV={key[2]} (any AO storage, either f16 rounding)" if key[0] == "any" else str(key))
    print(f"== {title}: {len(by_col[key])} kernels, {sum(n for _, n in by_col[key])} instructions")
    for n, k in sorted(by_col[key]):
        print(f"   {k:6d}  {n}")
print(f"== format-independent (atlas rebuild, debug views, composite, self-tests): {len(other)} kernels")
for n, k in sorted(other):
    print(f"   {k:6d}  {n}")
