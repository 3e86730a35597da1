"""Compile-time properties the performance design rests on, checked without a GPU: no kernel
spills or uses scratch, and the hot kernels keep the occupancy their LDS / VGPR budgets were cut for -- so a
compiler bump or an innocent edit cannot silently cost a wave per SIMD.  (tools/kernel_resources.py writes the
full table and the render-loop ISA into profiles/.)"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def rows():
    import shutil
    if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        pytest.skip("no hipcc here: the compile-time resource table cannot be produced")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_resources.py"), "--json"],
                         capture_output=True, text=True, check=True, cwd=ROOT, timeout=900)
    table = json.loads(out.stdout)
    assert len(table) > 100
    return {r["name"]: r for r in table}


def test_no_kernel_spills_or_uses_scratch(rows):
    # (SGPR spills go to VGPR lanes, not to memory: a few in the render kernels' term-constant tables are fine)
    bad = [n for n, r in rows.items()
           if int(r["ScratchSize [bytes/lane]"]) or int(r["VGPRs Spill"]) or r["Dynamic Stack"] != "False"]
    assert not bad, bad


# kernel (R8, RTZ, exact division = what bench.py times): (waves per SIMD, max VGPRs, max LDS bytes per workgroup)
HOT = {
    "render_kernel<0, false, 0, false>": (8, 64, 40960),                                   # 4 workgroups of 8 waves per CU
    "render_with_composite_kernel<0, false, 0>": (8, 64, 40960),
    "upsample_final_kernel<0, false, 0, true>": (7, 72, 163840 // 7),                       # seven 256-thread workgroups per CU
    "upsample_final_with_next_downsample_kernel<0, false, 0>": (7, 72, 163840 // 7),
    "upsample_kernel<0, false, 0>": (8, 64, 163840 // 8),
    "upsample_two_level_kernel<0, false, 0>": (8, 64, 163840 // 8),                       # compiled for 8 waves per SIMD: 4080 workgroups = 1.99 rounds of the slots
    "downsample_kernel<true, 0, 2>": (8, 64, 0),
}


@pytest.mark.parametrize("name", sorted(HOT))
def test_hot_kernels_keep_their_occupancy(rows, name):
    waves, vgprs, lds = HOT[name]
    r = rows[name]
    assert int(r["Occupancy [waves/SIMD]"]) >= waves, r
    assert int(r["VGPRs"]) <= vgprs and int(r["AGPRs"]) == 0, r
    assert int(r["LDS Size [bytes/block]"]) <= lds, r


def test_fp16_ao_variants_of_the_hot_kernels_keep_the_same_occupancy(rows):
    for name, (waves, vgprs, lds) in HOT.items():
        if "<0, " not in name:
            continue
        r = rows[name.replace("<0, ", "<1, ", 1)]          # AOFMT = F16 (BASELINE config 5)
        assert int(r["Occupancy [waves/SIMD]"]) >= waves and int(r["VGPRs"]) <= vgprs and int(r["LDS Size [bytes/block]"]) <= lds, r
