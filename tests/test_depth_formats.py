"""Input-side depth formats: the downsample kernel decodes the depth buffer's
storage format on load instead of running the reference's depth-copy blit (Blit.shader pass 0)."""
import ctypes as C

import numpy as np
import pytest

from miniengineao_amd import synth
from tests import helpers as H

FORMATS = {"f32": 0, "unorm16": 1, "unorm24": 2, "f16": 3}


def test_unorm_decode_is_exact_for_every_code(oracle):
    """v / (2^n - 1), correctly rounded, for all 2^16 codes; for 2^24 the kernel's 3-operation
    sequence (q = v*r; q += fma(-D, q, v) * r) is compared with IEEE division in C below."""
    L = oracle.lib()
    L.meao_oracle_decode_depth.restype = C.c_float
    L.meao_oracle_decode_depth.argtypes = [C.c_void_p, C.c_uint64, C.c_int32]
    codes = np.arange(65536, dtype=np.uint16)
    want = (codes.astype(np.float64) / 65535.0).astype(np.float32)      # double then one rounding: exact for 16 bits
    got = np.array([L.meao_oracle_decode_depth(codes.ctypes.data, i, 1) for i in range(0, 65536, 97)], np.float32)
    assert np.array_equal(got, want[::97])
    words = np.array([0, 1, 0xFFFFFF, 0xA5FFFFFF, 0x12800000, 0xFF000001], np.uint32)
    got = [L.meao_oracle_decode_depth(words.ctypes.data, i, 2) for i in range(len(words))]
    assert got == [0.0, np.float32(1 / 16777215.0), 1.0, 1.0, np.float32(0x800000 / 16777215.0), np.float32(1 / 16777215.0)]


def test_unorm_fma_sequence_matches_division(tmp_path):
    src = tmp_path / "t.c"
    src.write_text(r"""
#include <math.h>
#include <stdio.h>
int main(void) {
    for (int n = 16; n <= 24; n += 8) {
        const float D = (float)((1u << n) - 1u), r = 1.0f / D;
        for (unsigned v = 0; v < (1u << n); v++) {
            const float f = (float)v, q = f * r, e = fmaf(-D, q, f);
            if (fmaf(e, r, q) != f / D) { printf("mismatch n=%d v=%u\n", n, v); return 1; }
        }
    }
    puts("ok");
    return 0;
}
""")
    import subprocess
    exe = tmp_path / "t"
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-mfma", str(src), "-o", str(exe), "-lm"], check=True)
    assert subprocess.run([str(exe)], capture_output=True, text=True).stdout.strip() == "ok"


@pytest.mark.parametrize("fmt", sorted(FORMATS))
@pytest.mark.parametrize("reversed_z", [True, False])
def test_two_restatements_agree_for_every_depth_format(oracle, fmt, reversed_z):
    w, h = 75, 46
    cam = synth.Camera(reversed_z=reversed_z)
    raw = synth.occluder_field(w, h, seed=31, cam=cam)
    raw[3:9, 40:] = 0.0 if reversed_z else 1.0                      # sky: code 0 / all-ones
    depth = oracle.encode_depth(raw, FORMATS[fmt])
    s = H.settings(oracle, w, h, cam=cam, depth_format=FORMATS[fmt])
    a, b = oracle.run(depth, s), oracle.run(depth, s, emulate_hlsl=True)
    for name in a:
        assert np.array_equal(a[name], b[name]), name
    if fmt == "f32":
        return
    # decoding first and feeding float32 is the reference's two-step path (blit, then DS1)
    if fmt == "unorm16":
        decoded = (depth.astype(np.float64) / 65535.0).astype(np.float32)
    elif fmt == "unorm24":
        decoded = ((depth & np.uint32(0xFFFFFF)).astype(np.float64) / 16777215.0).astype(np.float32)
    else:
        decoded = depth.view(np.float16).astype(np.float32)
    c = oracle.run(decoded, H.settings(oracle, w, h, cam=cam))
    for name in a:
        assert np.array_equal(a[name], c[name]), name


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", sorted(FORMATS))
@pytest.mark.parametrize("w,h", [(256, 144), (131, 77)])
def test_gpu_depth_formats(oracle, fmt, w, h):
    raw = synth.occluder_field(w, h, seed=77)
    raw[5:15, w // 2:] = 0.0                                        # sky (reversed Z)
    depth = oracle.encode_depth(raw, FORMATS[fmt])
    s = H.settings(oracle, w, h, depth_format=FORMATS[fmt])
    want = oracle.run(depth, s)
    from miniengineao_amd import AmbientOcclusion
    ao = AmbientOcclusion(w, h, depth_format=FORMATS[fmt], near_clip=s.near_clip, far_clip=s.far_clip,
                          projection00=s.proj00, reversed_z=True, max_batch=2)
    try:
        outs = ao.render_batch([depth, depth])
        assert np.array_equal(outs[0], want["result"]) and np.array_equal(outs[1], want["result"])
        for i in (1, 2, 5, 6, 10, 14):
            assert np.array_equal(ao.debug_buffer(i, frame=1), want[H.NAMES[i]]), H.NAMES[i]
    finally:
        ao.close()
