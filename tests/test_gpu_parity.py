"""GPU parity: the HIP path through the C ABI vs the CPU oracle, bit-exact.

Every AO texel is integer-valued storage (UNORM8 or f16 bit patterns) and every depth
intermediate is an exact f32/f16 bit pattern, so the bar is equality of the raw arrays.
(north_star's "within 1 ULP fp16" is met with margin: tolerance here is 0 ULP.)
"""
import numpy as np
import pytest

from miniengineao_amd import synth
from tests import helpers as H

pytestmark = pytest.mark.gpu

SIZES = [(67, 45), (130, 70), (256, 256), (33, 17), (1, 1), (5, 300), (300, 5), (64, 64),
         (129, 65), (640, 360)]


def _compare_all(O, s, depth, ao=None):
    want = O.run(depth, s)
    own = ao is None
    ao = ao or H.component(s)
    try:
        got = ao.render(depth)
        assert np.array_equal(got, want["result"]), H.diff_report("result", got, want["result"])
        for i in H.valid_debug_ids(s.num_levels, s.hq_levels):
            g = ao.debug_buffer(i)
            w = want[H.NAMES[i]]
            assert g.shape == w.shape and g.dtype == w.dtype, (i, g.shape, w.shape, g.dtype, w.dtype)
            assert np.array_equal(g, w), H.diff_report(H.NAMES[i], g, w)
    finally:
        if own:
            ao.close()


@pytest.mark.parametrize("w,h", SIZES)
@pytest.mark.parametrize("kind", ["S1", "S2"])
def test_all_17_buffers_match_oracle(oracle, w, h, kind):
    s = H.settings(oracle, w, h)
    _compare_all(oracle, s, synth.make(kind, w, h))


@pytest.mark.parametrize("ao_format", [0, 1])
@pytest.mark.parametrize("f16_rounding", [0, 1])
@pytest.mark.parametrize("num_levels", [1, 2, 3, 4])
def test_modes(oracle, ao_format, f16_rounding, num_levels):
    w, h = 203, 117
    s = H.settings(oracle, w, h, ao_format=ao_format, f16_rounding=f16_rounding, num_levels=num_levels)
    _compare_all(oracle, s, synth.make("S2", w, h, seed=77))


@pytest.mark.parametrize("variant", [
    dict(hq_levels=4), dict(hq_levels=1), dict(hq_levels=2, num_levels=3), dict(sample_set=1),
    dict(single_pass_stereo=True), dict(hq_levels=4, sample_set=1, single_pass_stereo=True, ao_format=1),
    dict(hq_levels=3, sample_set=1, f16_rounding=1, ao_format=1), dict(hq_levels=1, num_levels=1),
])
@pytest.mark.parametrize("w,h", [(203, 117), (256, 128), (61, 190)])
def test_variants(oracle, variant, w, h):
    """Render.main (wide) + Upsample.main_premin*, SAMPLE_EXHAUSTIVELY, single-pass stereo:
    every buffer incl. OcclusionHQ<k> bit-exact against the oracle."""
    s = H.settings(oracle, w, h, **variant)
    depth = synth.make("S2", w, h, seed=31)
    depth[h // 4: h // 4 + 20, w // 3: w // 3 + 50] = 0.0           # sky texels
    _compare_all(oracle, s, depth)


def test_variant_buffers_are_absent_without_the_variant(oracle):
    s = H.settings(oracle, 96, 64, hq_levels=2)
    ao = H.component(s)
    try:
        ao.render(synth.make("S1", 96, 64))
        assert ao.debug_buffer(21).shape == (4, 6) and ao.debug_buffer(20).shape == (8, 12)
        for i in (18, 19):                                         # levels 1, 2 have no Render.main pass
            with pytest.raises(Exception, match="hq_levels"):
                ao.debug_buffer(i)
    finally:
        ao.close()


@pytest.mark.parametrize("params", [
    dict(intensity=0.0), dict(intensity=2.0), dict(thickness_modifier=10.0),
    dict(blur_tolerance=-1.0), dict(blur_tolerance=-8.0), dict(upsample_tolerance=-1.0),
    dict(noise_filter_tolerance=-8.0), dict(intensity=1.1, thickness_modifier=3.0, blur_tolerance=-3.0,
                                            upsample_tolerance=-5.0, noise_filter_tolerance=-2.0),
])
def test_parameter_sweep(oracle, params):
    w, h = 161, 99
    s = H.settings(oracle, w, h, **params)
    _compare_all(oracle, s, synth.make("S2", w, h, seed=5))


def test_conventional_z_and_sponza_camera(oracle):
    w, h = 240, 135
    cam = synth.Camera(near=0.01, far=100.0, fov_y_deg=30.0, reversed_z=False)
    lin = np.clip(synth.linear01_to_raw(np.full((h, w), 0.5), cam), 0, 1)  # placeholder plane
    depth = synth.linear01_to_raw(0.2 + 0.6 * np.random.default_rng(3).random((h, w)), cam)
    s = H.settings(oracle, w, h, cam=cam, intensity=1.1)
    _compare_all(oracle, s, depth)
    _compare_all(oracle, s, lin)


@pytest.mark.parametrize("f16_rounding", [0, 1])
@pytest.mark.parametrize("reversed_z", [True, False])
def test_sky_texels(oracle, f16_rounding, reversed_z):
    """Sky (raw depth 0 reversed / 1 conventional -> 1e5, overflows f16) next to geometry."""
    w, h = 150, 90
    cam = synth.Camera(reversed_z=reversed_z)
    depth = synth.occluder_field(w, h, seed=9, cam=cam)
    sky = np.float32(0.0 if reversed_z else 1.0)
    depth[:, : w // 3] = sky
    depth[h // 2:, w // 2:] = sky
    s = H.settings(oracle, w, h, cam=cam, f16_rounding=f16_rounding)
    _compare_all(oracle, s, depth)


def test_resize_and_property_change(oracle):
    s = H.settings(oracle, 96, 64)
    ao = H.component(s)
    try:
        _compare_all(oracle, s, synth.make("S2", 96, 64), ao)
        ao.intensity = 1.7
        ao.thicknessModifier = 2.0
        s2 = H.settings(oracle, 96, 64, intensity=1.7, thickness_modifier=2.0)
        _compare_all(oracle, s2, synth.make("S2", 96, 64), ao)
        ao.resize(145, 77)
        ao.projection00 = synth.DEFAULT_CAMERA.proj00(145, 77)
        s3 = H.settings(oracle, 145, 77, intensity=1.7, thickness_modifier=2.0)
        _compare_all(oracle, s3, synth.make("S2", 145, 77), ao)
    finally:
        ao.close()


def test_batch_matches_per_frame(oracle):
    w, h, n = 200, 120, 5
    s = H.settings(oracle, w, h)
    depths = [synth.make("S2", w, h, seed=100 + f) for f in range(n)]
    ao = H.component(s, max_batch=8)
    try:
        outs = ao.render_batch(depths)
        for f in range(n):
            want = oracle.run(depths[f], s)
            assert np.array_equal(outs[f], want["result"]), f
            for i in (2, 5, 7, 10, 13, 14, 16):
                assert np.array_equal(ao.debug_buffer(i, frame=f), want[H.NAMES[i]]), (f, i)
    finally:
        ao.close()


def test_largest_batch(oracle):
    """MEAO_MAX_BATCH = 64 frames through one launch per pass, incl. the pipelined path."""
    import torch
    from miniengineao_amd import _lib as L
    w, h, n = 136, 72, L.MAX_BATCH
    s = H.settings(oracle, w, h)
    depths = [synth.make("S2", w, h, seed=300 + f) for f in range(n)]
    wants = [oracle.run(d, s, result_only=True)["result"] for d in depths]
    ao = H.component(s, max_batch=n)
    try:
        outs = ao.render_batch(depths)
        assert all(np.array_equal(outs[f], wants[f]) for f in range(n))
        d_in = [torch.from_numpy(d).cuda() for d in depths]
        d_out = [torch.zeros((h, w), dtype=torch.uint8, device="cuda") for _ in range(n)]
        pin, pout = [t.data_ptr() for t in d_in], [t.data_ptr() for t in d_out]
        for _ in range(2):
            ao.prefetch_device(pin)
            ao.execute_device(pin, pout)
        ao.synchronize()
        assert all(np.array_equal(d_out[f].cpu().numpy(), wants[f]) for f in range(n))
        assert np.array_equal(ao.debug_buffer(2, frame=n - 1), oracle.run(depths[n - 1], s)["low_depth1"])
    finally:
        ao.close()


@pytest.mark.parametrize("which", [0, 1, 2, 3, 4, 5, 6, 7])
def test_hardware_conversions_exhaustive(oracle, which):
    """All 2^32 f32 -> f16 inputs (both rounding modes), all 256 UNORM8 and all 65536 f16
    decodes: hardware conversion == the bit-level model the oracle uses; 4-6: the exact-division sequences; 7: the UNORM8
    bilateral result from uncorrected reciprocals (bilateral_upsample_r8) == the code of the exact chain."""
    s = H.settings(oracle, 16, 16)
    ao = H.component(s)
    try:
        assert ao.selftest(which) == 0
    finally:
        ao.close()


@pytest.mark.parametrize("w,h,kind", [(1920, 1080, "S3"), (1920, 1080, "S2")])
def test_1080p_full_frame(oracle, w, h, kind):
    cam = synth.SPONZA_CAMERA if kind == "S3" else synth.DEFAULT_CAMERA
    depth = synth.atrium(w, h) if kind == "S3" else synth.make(kind, w, h)
    s = H.settings(oracle, w, h, cam=cam, intensity=1.1 if kind == "S3" else 1.0)
    want = oracle.run(depth, s, nthreads=8, result_only=True)["result"]
    ao = H.component(s)
    try:
        got = ao.render(depth)
    finally:
        ao.close()
    assert np.array_equal(got, want), H.diff_report("result", got, want)


def test_4k_full_frame_and_properties(oracle):
    """BASELINE config 3 at full size: bit-exact vs the oracle (threaded), plus the
    size-independent properties: constant depth -> all 255; intensity 0 -> all 255."""
    w, h = 3840, 2160
    depth = synth.make("S2", w, h)
    s = H.settings(oracle, w, h)
    want = oracle.run(depth, s, nthreads=8, result_only=True)["result"]
    ao = H.component(s)
    try:
        got = ao.render(depth)
        assert np.array_equal(got, want), H.diff_report("result", got, want)
        flat = synth.linear01_to_raw(np.full((h, w), 0.37))
        assert (ao.render(flat) == 255).all()          # W, H multiples of 64: no padding texels
        ao.intensity = 0.0
        assert (ao.render(depth) == 255).all()
    finally:
        ao.close()
