"""The `_debug` 1..17 views (PushDebugBlitCommands AO.cs:787-820)."""
import numpy as np
import pytest

from miniengineao_amd import synth
from tests import helpers as H


def test_debug_view_oracle_geometry(oracle):
    """Tiled views are the 4x4 grid of slices (Blit.shader:136-155); 2D views are nearest-texel
    magnifications; id 17 is the result itself."""
    w, h = 128, 64
    s = H.settings(oracle, w, h)
    bufs = oracle.run(synth.make("S2", w, h), s)
    assert np.array_equal(oracle.debug_view(bufs, 17, s), bufs["result"])
    v = oracle.debug_view(bufs, 14, s)                       # Combined1 is exactly half resolution
    assert np.array_equal(v[::2, ::2], bufs["combined1"]) and np.array_equal(v[1::2, 1::2], bufs["combined1"])
    t = oracle.debug_view(bufs, 6, s)                        # TiledDepth1: 16 x 8 slices, shown 32 x 16 each
    atlas = oracle.f16_bits_to_f32(bufs["tiled_depth1"])
    for sl in (0, 5, 15):
        tile = t[(sl >> 2) * 16:(sl >> 2) * 16 + 16, (sl & 3) * 32:(sl & 3) * 32 + 32]
        want = np.clip(atlas[sl], 0, 1)
        assert np.array_equal(tile[::2, ::2], np.floor(want * np.float32(255) + np.float32(0.5)).astype(np.uint8))


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,ao_format", [(128, 64, 0), (203, 117, 0), (203, 117, 1)])
def test_gpu_debug_views_match_oracle(oracle, w, h, ao_format):
    s = H.settings(oracle, w, h, ao_format=ao_format)
    depth = synth.make("S2", w, h, seed=9)
    bufs = oracle.run(depth, s)
    ao = H.component(s)
    try:
        ao.render(depth)
        for i in range(1, 18):
            got, want = ao.debug_view(i), oracle.debug_view(bufs, i, s)
            assert np.array_equal(got, want), (i, H.diff_report(H.NAMES[i], got, want))
    finally:
        ao.close()
