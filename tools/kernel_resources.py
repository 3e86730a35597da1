#!/usr/bin/env python3
"""Compile-time evidence for the numbers DESIGN.md quotes: per-kernel VGPR / SGPR / LDS / scratch /
occupancy of every shipped gfx950 kernel (hipcc -Rpass-analysis=kernel-resource-usage) and the ISA of the
render texel loop (one sample pair).  Runs without a GPU (hipcc cross-compiles).

    python tools/kernel_resources.py [--out profiles/r03_kernel_resources.txt] [--isa profiles/r03_render_loop_isa.txt]
    python tools/kernel_resources.py --json          # machine-readable table on stdout (tests/test_kernel_resources.py)
"""
from __future__ import annotations

import argparse
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from miniengineao_amd import build  # noqa: E402

SRC = os.path.join(ROOT, "miniengineao_amd", "csrc", "meao_kernels.hip")


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    except OSError:
        return list(names)
    return out.stdout.splitlines() if out.returncode == 0 else list(names)


def short(name: str) -> str:
    name = name.replace("meao::(anonymous namespace)::", "")
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)


def compile_device(extra_flags=(), want_asm=True):
    """Returns (rows, asm_text); rows: one dict per kernel."""
    flags = [f for f in build.FLAGS if f not in ("-shared", "-fPIC", "-fvisibility=hidden")]
    with tempfile.TemporaryDirectory() as tmp:
        asm = os.path.join(tmp, "k.s")
        cmd = [build.hipcc(), *flags, *extra_flags, f"-I{os.path.join(ROOT, 'include')}", "--cuda-device-only", "-S",
               "-Rpass-analysis=kernel-resource-usage", SRC, "-o", asm]
        proc = subprocess.run(cmd, capture_output=True, text=True)
        if proc.returncode != 0:
            raise RuntimeError(proc.stderr[-4000:])
        text = open(asm).read() if want_asm else ""
    rows, cur = [], None
    for line in proc.stderr.splitlines():
        m = re.search(r"remark:\s+(.*?) \[-Rpass-analysis", line)
        if not m:
            continue
        body = m.group(1).strip()
        if body.startswith("Function Name:"):
            cur = {"mangled": body.split(":", 1)[1].strip()}
            rows.append(cur)
        elif cur is not None and ":" in body:
            k, v = body.rsplit(":", 1)
            cur[k.strip()] = v.strip()
    names = demangle([r["mangled"] for r in rows])
    for r, n in zip(rows, names):
        r["name"] = short(n)
    return rows, text


def table(rows) -> str:
    cols = [("VGPRs", "VGPR"), ("AGPRs", "AGPR"), ("TotalSGPRs", "SGPR"), ("LDS Size [bytes/block]", "LDS B"),
            ("ScratchSize [bytes/lane]", "scratch"), ("VGPRs Spill", "vspill"), ("SGPRs Spill", "sspill"),
            ("Occupancy [waves/SIMD]", "waves/SIMD")]
    lines = ["%-5s %-5s %-5s %-7s %-8s %-7s %-7s %-11s kernel" % tuple(c[1] for c in cols)]
    for r in sorted(rows, key=lambda r: r["name"]):
        lines.append("%-5s %-5s %-5s %-7s %-8s %-7s %-7s %-11s %s" % (*[r.get(c[0], "?") for c in cols], r["name"]))
    return "\n".join(lines)


def render_loop_isa(asm: str) -> str:
    """The texel loop of the R8 / RTZ / exact-division render kernel: from the first hand-issued ds_read_b64 to
    the store, with a per-class instruction count of one steady-state sample pair."""
    m = re.search(r"^(_ZN4meao\w*13render_kernelILi0ELb0ELi0ELb0E\w*):[^\n]*\n(.*?)^\.Lfunc_end", asm, re.S | re.M)
    if not m:
        return "render_kernel<R8, RTZ, exact, checker> not found in the assembly\n"
    body = m.group(2).splitlines()
    idx = [i for i, l in enumerate(body) if "ds_read_b64" in l]
    if len(idx) < 8:
        return "no hand-pipelined ds_read_b64 sequence found\n"
    # a steady-state pair: between the 5th and 6th "s_waitcnt lgkmcnt(2)" of the loop
    waits = [i for i, l in enumerate(body) if re.search(r"s_waitcnt lgkmcnt\(2\)", l)]
    out = []
    # ... that is not the first pair of a term (those also carry the term's v_pk_mul of invThickness * invDepth)
    # and not the last one either (those add the term's sum and weighted accumulation): the segment with the fewest VALU instructions
    def valu_count(k):
        return sum(1 for l in body[waits[k]:waits[k + 1]] if l.strip().startswith("v_"))
    cands = [k for k in range(4, len(waits) - 1) if not any("v_pk_mul" in l for l in body[waits[k]:waits[k + 1]])]
    pick = min(cands, key=valu_count) if cands else None
    if pick is not None:
        a, b = waits[pick], waits[pick + 1]
        seg = [l.split(";")[0].rstrip() for l in body[a:b] if l.strip() and not l.strip().startswith((";", "."))]
        out.append(f"; one steady-state sample pair of the checker set (between two 's_waitcnt lgkmcnt(2)'), {len(seg)} instructions:")
        out += seg
        ops = [l.split()[0] for l in seg]
        valu = [o for o in ops if o.startswith("v_")]
        out.append(f"; VALU instructions in the pair (two texels per lane): {len(valu)} = {len(valu) // 2} per texel -> "
                   + ", ".join(f"{o} x{valu.count(o)}" for o in sorted(set(valu))))
    out.append("")
    out.append(f"; whole texel loop: {idx[0]}..{idx[-1]} of {len(body)} lines; ds_read_b64 count {len(idx)}, "
               f"lgkmcnt(2) waits {len(waits)}")
    return "\n".join(out) + "\n"


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--out")
    ap.add_argument("--isa")
    ap.add_argument("--json", action="store_true")
    ap.add_argument("-D", action="append", default=[], help="extra -D flags (variants)")
    a = ap.parse_args()
    rows, asm = compile_device(["-D" + d for d in a.D], want_asm=bool(a.isa) or not a.json)
    if a.json:
        print(json.dumps([{k: v for k, v in r.items() if k != "mangled"} for r in rows]))
        return 0
    head = ("# hipcc " + " ".join(f for f in build.FLAGS if f.startswith(("-O", "-f", "--off"))) +
            " -Rpass-analysis=kernel-resource-usage (ROCm " + os.path.realpath("/opt/rocm").rsplit("-", 1)[-1] + ")\n"
            "# gfx950: 512 VGPRs per SIMD lane, allocation granule 8 -> waves/SIMD = floor(512 / ceil8(VGPR)), max 8;\n"
            "# LDS 160 KB per CU.  Template arguments: <AOFMT (0 = R8, 1 = F16), RTNE, [FINAL,] DIV (0 = exact rcp, 1 = IEEE, 2 = fast), ...>\n")
    txt = head + table(rows) + "\n"
    if a.out:
        open(a.out, "w").write(txt)
    else:
        print(txt)
    if a.isa:
        open(a.isa, "w").write(render_loop_isa(asm))
    return 0


if __name__ == "__main__":
    sys.exit(main())
