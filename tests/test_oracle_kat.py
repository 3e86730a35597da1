"""Oracle self-tests (CPU).  The reference ships no goldens, so the oracle is pinned by
(a) known-answer values derived from the reference source, (b) bit-for-bit
agreement of two independently structured restatements, (c) analytical properties."""
import ctypes as C

import numpy as np
import pytest

from miniengineao_amd import synth
from tests import helpers as H


def bits(x):
    return int(np.float32(x).view(np.uint32))


def test_sample_thickness_bits(oracle):
    # AO.cs:577-590 evaluated in float; bit patterns from SURVEY.md 8a row a6
    want = [0x3f7ad3e7, 0x3f6aa0bc, 0x3f4ccccd, 0x3f199999, 0x3f758bec, 0x3f64f92e,
            0x3f464bf7, 0x3f10d0c2, 0x3f531a5e, 0x3f315cac, 0x3ee4f92b, 0x3f077664]
    assert [bits(v) for v in oracle.sample_thickness()] == want


def test_sample_weights_kat(oracle):
    s = H.settings(oracle, 3840, 2160)
    w = np.array(list(oracle.render_constants(s, 1).sample_weight), np.float32)
    want = np.array([0, .146103054, 0, .0956468955, .152902141, 0, .246959224, 0, .13145408, 0,
                     .142581955, .0843526348], np.float32)
    assert np.array_equal(w, want)
    total = np.float32(0)
    for v in w:
        total = np.float32(total + v)
    assert bits(total) == 0x3f7ffffe            # normalised weights sum to 1 - 2ulp in fp32
    for level in (2, 3, 4):                      # weights do not depend on the level
        assert np.array_equal(np.array(list(oracle.render_constants(s, level).sample_weight), np.float32), w)


def test_render_constants_structure(oracle):
    s = H.settings(oracle, 3840, 2160, thickness_modifier=4.0, intensity=1.3)
    thick = oracle.sample_thickness()
    for level, sw in ((1, 480), (2, 240), (3, 120), (4, 60)):
        rc = oracle.render_constants(s, level)
        tan_half = np.float32(1.0) / np.float32(s.proj00)
        mult = np.float32(np.float32(np.float32(2.0) * tan_half) * np.float32(10.0)) / np.float32(sw)
        inv = np.float32(1.0) / mult
        assert np.array_equal(np.array(list(rc.inv_thickness), np.float32), (inv / thick).astype(np.float32))
        assert rc.reject_fadeoff == np.float32(-0.25) and rc.intensity == np.float32(1.3)
        assert rc.inv_slice_dim[0] == np.float32(1.0) / np.float32(sw)


@pytest.mark.parametrize("w,steps,blur", [
    (3840, [1, 2, 4, 8], [.999949813, .999899507, .999799013, .999598205]),
    (1920, [2, 4, 8, 16], [.999899507, .999799013, .999598205, .999196351])])
def test_upsample_constants_kat(oracle, w, steps, blur):
    s = H.settings(oracle, w, w * 9 // 16)
    for low_level in (1, 2, 3, 4):
        u = oracle.upsample_constants(s, low_level)
        assert u.step_size == steps[low_level - 1]             # 1920 / lowRes.width (AO.cs:760)
        assert u.blur_tolerance == np.float32(blur[low_level - 1])
        assert u.upsample_tolerance == np.float32(1e-12)
        assert u.noise_filter_strength == np.float32(1.0)


def test_zbuffer_params(oracle):
    s = H.settings(oracle, 64, 64)
    fpn = np.float32(100.0) / np.float32(0.1)
    assert oracle.zbuffer_params(s)[:2] == [fpn - np.float32(1), 1.0]
    s.reversed_z = False
    assert oracle.zbuffer_params(s)[:2] == [np.float32(1) - fpn, fpn]


def test_f16_conversions(oracle):
    L = oracle.lib()
    rng = np.random.default_rng(1)
    vals = np.concatenate([
        rng.standard_normal(20000).astype(np.float32) * np.float32(10.0) ** rng.integers(-9, 6, 20000).astype(np.float32),
        np.array([0.0, -0.0, 1.0, 65504.0, 65519.9, 65520.0, 1e5, -1e5, np.inf, -np.inf, 6.1e-5, 6.0e-8,
                  2.98e-8, 2.99e-8, 1e-10, 0.33325195, 0.3334, 1 - 2 ** -12, 1 + 2 ** -11], np.float32)])
    with np.errstate(over="ignore"):
        rtne_ref = vals.astype(np.float16).view(np.uint16)     # numpy converts RTNE
    for v, ref in zip(vals, rtne_ref):
        assert L.meao_oracle_f32_to_f16(float(v), oracle.F16_RTNE) == int(ref), v
        rtz = L.meao_oracle_f32_to_f16(float(v), oracle.F16_RTZ)
        back = np.uint16(rtz).view(np.float16).astype(np.float64)
        if np.isfinite(v):
            assert abs(back) <= abs(float(v)), v                # toward zero
            nxt = np.uint16(rtz + 1).view(np.float16).astype(np.float64) if (rtz & 0x7fff) < 0x7bff else np.inf
            assert abs(nxt) > abs(float(v)) or abs(float(v)) >= 65504.0, v   # and tight
    assert L.meao_oracle_f32_to_f16(1e5, oracle.F16_RTZ) == 0x7bff            # sky clamps to 65504
    assert L.meao_oracle_f32_to_f16(1e5, oracle.F16_RTNE) == 0x7c00           # or overflows to +inf
    every = np.arange(65536, dtype=np.uint16)
    ref32 = every.view(np.float16).astype(np.float32)
    for b in (0, 1, 0x3ff, 0x400, 0x3c00, 0x7bff, 0x7c00, 0x8001, 0xfbff, 0x1234, 0xabcd):
        assert bits(L.meao_oracle_f16_to_f32(b)) == bits(ref32[b])


def test_unorm8_conversions(oracle):
    L = oracle.lib()
    for n in range(256):
        f = L.meao_oracle_unorm8_to_f32(n)
        assert f == np.float32(n) / np.float32(255)
        assert L.meao_oracle_f32_to_unorm8(f) == n
    assert L.meao_oracle_f32_to_unorm8(float("nan")) == 0
    assert L.meao_oracle_f32_to_unorm8(-3.0) == 0 and L.meao_oracle_f32_to_unorm8(7.0) == 255
    assert L.meao_oracle_f32_to_unorm8(0.5) == 128          # 127.5 + 0.5 truncates to 128


ODD_SIZES = [(67, 45), (130, 70), (33, 17), (1, 1), (2, 3), (5, 300), (300, 5), (129, 65), (255, 257)]


@pytest.mark.parametrize("w,h", ODD_SIZES)
def test_two_restatements_agree_bit_for_bit(oracle, w, h):
    """Gather-form oracle vs literal thread-group/LDS emulation: all 17 buffers identical."""
    for kind in ("S1", "S2"):
        depth = synth.make(kind, w, h, seed=w * 1000 + h)
        s = H.settings(oracle, w, h)
        a, b = oracle.run(depth, s), oracle.run(depth, s, emulate_hlsl=True)
        for name in a:
            assert np.array_equal(a[name], b[name]), H.diff_report(name, a[name], b[name])


@pytest.mark.parametrize("ao_format", [0, 1])
@pytest.mark.parametrize("f16_rounding", [0, 1])
@pytest.mark.parametrize("num_levels", [1, 2, 3, 4])
@pytest.mark.parametrize("reversed_z", [True, False])
def test_two_restatements_agree_in_every_mode(oracle, ao_format, f16_rounding, num_levels, reversed_z):
    w, h = 97, 61
    cam = synth.Camera(reversed_z=reversed_z)
    depth = synth.occluder_field(w, h, seed=4242, cam=cam)
    depth[5:20, 60:] = 0.0 if reversed_z else 1.0               # sky block (1e5, overflows f16)
    s = H.settings(oracle, w, h, cam=cam, ao_format=ao_format, f16_rounding=f16_rounding,
                   num_levels=num_levels, intensity=1.4, thickness_modifier=2.0, blur_tolerance=-3.0,
                   upsample_tolerance=-6.0, noise_filter_tolerance=-1.0)
    a, b = oracle.run(depth, s), oracle.run(depth, s, emulate_hlsl=True)
    for i in H.valid_debug_ids(num_levels):
        name = H.NAMES[i]
        assert np.array_equal(a[name], b[name]), H.diff_report(name, a[name], b[name])


def test_thread_count_does_not_change_results(oracle):
    w, h = 321, 200
    depth = synth.make("S2", w, h)
    s = H.settings(oracle, w, h)
    a, b = oracle.run(depth, s, nthreads=1), oracle.run(depth, s, nthreads=5)
    for name in a:
        assert np.array_equal(a[name], b[name]), name


@pytest.mark.parametrize("w,h", [(64, 64), (256, 128), (192, 320)])
@pytest.mark.parametrize("ao_format", [0, 1])
def test_constant_depth_gives_no_occlusion(oracle, w, h, ao_format):
    """SURVEY 8c KAT (2): constant depth, W and H multiples of 64 (no padding texels) -> every
    sample pair contributes 0.5+0.5, ao = sum of weights ~ 1, blur and bilateral of a constant
    are the identity -> all 255 (R8) / 1.0 minus at most one RTZ truncation per pass (F16)."""
    depth = synth.linear01_to_raw(np.full((h, w), 0.3))
    s = H.settings(oracle, w, h, ao_format=ao_format, thickness_modifier=3.0)
    out = oracle.run(depth, s)["result"]
    if ao_format == 0:
        assert (out == 255).all()
    else:
        assert np.abs(oracle.f16_bits_to_f32(out) - 1.0).max() <= 6 * 2.0 ** -11


def test_intensity_zero_is_identity(oracle):
    """SURVEY 8c KAT (3): lerp(1, ao, 0) = 1 regardless of depth; upsample multiplies ones."""
    w, h = 150, 83
    s = H.settings(oracle, w, h, intensity=0.0)
    out = oracle.run(synth.make("S2", w, h), s)
    for name in ("occlusion1", "occlusion4", "combined2", "result"):
        assert (out[name] == 255).all(), name


def test_downsample_index_identities(oracle):
    """SURVEY 8c KAT (4): DSkx[i,j] = lin(k i, k j); atlas slice = (x&3) + 4(y&3); padding rule;
    the de-tile formula of Blit.shader:150-151 inverts the atlas."""
    w, h = 203, 117
    depth = synth.make("S2", w, h, seed=31)
    s = H.settings(oracle, w, h)
    out = oracle.run(depth, s)
    zp = oracle.zbuffer_params(s)
    # mad(ZP.x, d, ZP.y): the product of two f32 is exact in f64, the sum rounds once to f32
    den = (np.float64(zp[0]) * depth.astype(np.float64) + np.float64(zp[1])).astype(np.float32)
    lin32 = (np.float32(1.0) / den).astype(np.float32)
    assert np.array_equal(out["low_depth1"], lin32[::2, ::2])
    assert np.array_equal(out["low_depth2"], lin32[::4, ::4])
    assert np.array_equal(out["low_depth3"], lin32[::8, ::8])
    assert np.array_equal(out["low_depth4"], lin32[::16, ::16])
    L = oracle.lib()
    for k, pad in ((1, 1e5), (2, 1e5), (3, 0.0), (4, 0.0)):
        low, atlas = out[f"low_depth{k}"], out[f"tiled_depth{k}"]
        lh, lw = low.shape
        _, th, tw = atlas.shape
        padded = np.full((4 * th, 4 * tw), np.float32(pad), np.float32)
        padded[:lh, :lw] = low
        want = np.vectorize(lambda v: L.meao_oracle_f32_to_f16(float(v), 0), otypes=[np.uint16])(padded)
        for sl in range(16):
            assert np.array_equal(atlas[sl], want[(sl >> 2)::4, (sl & 3)::4]), (k, sl)
        # Blit.shader pass 4: uv*4 -> slice = floor(u4) + 4 floor(v4), inside-slice = frac
        grid = np.block([[atlas[r * 4 + c] for c in range(4)] for r in range(4)])
        assert grid.shape == (4 * th, 4 * tw)
        assert np.array_equal(grid[th:2 * th, 2 * tw:3 * tw], atlas[6])


def test_render_reads_only_its_own_slice(oracle):
    """SURVEY 8c KAT (5): Occlusion(X,Y) depends only on slice (X&3)+4(Y&3) within +-4 slice
    texels of (X>>2, Y>>2): changing one level-1 texel leaves all other slices untouched."""
    w, h = 160, 96
    depth = synth.make("S2", w, h, seed=8)
    s = H.settings(oracle, w, h, num_levels=1)
    base = oracle.run(depth, s)["occlusion1"]
    poked = depth.copy()
    poked[40, 82] = synth.linear01_to_raw(np.array([[0.9]]))[0, 0]   # level-1 texel (41, 20): slice (1, 0)
    out = oracle.run(poked, s)["occlusion1"]
    changed = np.argwhere(out != base)
    assert len(changed) > 0
    assert all((x & 3) == 1 and (y & 3) == 0 for y, x in changed)
    assert all(abs((x >> 2) - 10) <= 4 and abs((y >> 2) - 5) <= 4 for y, x in changed)


def test_step_edge_locality(oracle):
    """SURVEY 8c KAT (7): far from a depth step between two constant half-planes the output is
    exactly unoccluded; the farthest tap is 256 full-res px (level 4) + blur aprons."""
    w, h = 1536, 192
    lin = np.full((h, w), 0.2)
    lin[:, w // 2:] = 0.6
    s = H.settings(oracle, w, h)
    out = oracle.run(synth.linear01_to_raw(lin), s)["result"]
    reach = 256 + 16 * 6          # level-4 taps + 2-texel blur + bilateral footprint per level
    assert (out[:, : w // 2 - reach] == 255).all()
    assert (out[:, w // 2 + reach:] == 255).all()
    assert (out[:, w // 2 - 8: w // 2 + 8] < 255).any()
