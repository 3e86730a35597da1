"""Hostile depth inputs: NaN, +-inf, negative, > 1, denormal and denominator-zeroing raw depths, and
cameras whose near-plane texels flush to zero in the f16 depth mips.

The reference divides with IEEE '/' and never sanitises its input (Downsample1.compute:37-48,
Render.compute:140, Upsample.compute:67,179-182), so the oracle does the same and the HIP path
must match it bit for bit: the AO buffers exactly, the depth buffers up to the payload of NaNs
(x86 and gfx950 generate different quiet-NaN bit patterns).  The product detects such frames in its
downsample pass and runs the IEEE-division bodies of the later kernels for them (per frame)."""
import numpy as np
import pytest

from miniengineao_amd import synth
from tests import helpers as H


_nan_aware_equal = H.nan_aware_equal
hostile_frame = H.hostile_frame


def _compare(O, s, depth, ao=None, frame=0, got=None, debug=None):
    want = O.run(depth, s)
    own = ao is None
    ao = ao or H.component(s, debug=debug)
    try:
        if got is None:
            got = ao.render(depth)
        ok, bad = _nan_aware_equal(got, want["result"])
        assert ok, f"result: {int(bad.sum())} texels differ, first {tuple(np.argwhere(bad)[0])}"
        for i in H.valid_debug_ids(s.num_levels, s.hq_levels):
            g = ao.debug_buffer(i, frame)
            wv = want[H.NAMES[i]]
            ok, bad = _nan_aware_equal(g, wv)
            if not ok:
                at = tuple(np.argwhere(bad)[0])
                raise AssertionError(f"{H.NAMES[i]}: {int(bad.sum())} of {g.size} differ; first at {at}: "
                                     f"got {g[at]!r} want {wv[at]!r}")
    finally:
        if own:
            ao.close()


def test_oracle_restatements_agree_on_hostile_depth(oracle):
    """CPU: the gather-form oracle and the literal HLSL emulation agree on hostile frames too."""
    w, h = 67, 45
    s = H.settings(oracle, w, h)
    depth = hostile_frame(w, h, 5, density=0.03)
    a = oracle.run(depth, s)
    b = oracle.run(depth, s, emulate_hlsl=True)
    for name in a:
        ok, bad = _nan_aware_equal(a[name], b[name])
        assert ok, (name, int(bad.sum()))


@pytest.mark.gpu
@pytest.mark.parametrize("f16_rounding", [0, 1])
@pytest.mark.parametrize("ao_format", [0, 1])
@pytest.mark.parametrize("w,h,seed", [(203, 117, 1), (256, 128, 2), (67, 45, 3)])
def test_hostile_f32_depth_matches_oracle(oracle, w, h, seed, ao_format, f16_rounding):
    s = H.settings(oracle, w, h, ao_format=ao_format, f16_rounding=f16_rounding)
    _compare(oracle, s, hostile_frame(w, h, seed))


# Where a hostile texel sits decides who sees it: a texel of the LEVELS (even row and column) is found by the downsample pass, which
# flags the frame (IEEE-division bodies in every later kernel); any other texel is only ever linearized by the full-resolution
# upsample, which redoes the lane's texels with IEEE '/' (hi_depth_quad / the cold loop of upsample_tile<FINAL>), no frame flag.
PLACEMENTS = ("anywhere", "odd_texels_only", "level_texels_only")


def _placed(frame, clean, placement):
    """Keep the hostile texels of `frame` only at the wanted positions (the others come from the clean frame)."""
    if placement == "anywhere":
        return frame
    yy, xx = np.mgrid[0:frame.shape[0], 0:frame.shape[1]]
    level = ((yy & 1) == 0) & ((xx & 1) == 0)
    return np.where(level if placement == "level_texels_only" else ~level, frame, clean).astype(np.float32)


@pytest.mark.gpu
@pytest.mark.parametrize("placement", PLACEMENTS)
@pytest.mark.parametrize("kind", ["nan", "pinf", "ninf", "neg", "big", "huge", "nhuge", "denorm", "negzero",
                                  "zero_den", "tiny_den", "one", "zero"])
def test_each_hostile_value_alone(oracle, kind, placement):
    w, h = 130, 70
    s = H.settings(oracle, w, h)
    clean = synth.make("S2", w, h, seed=11)
    frame = _placed(hostile_frame(w, h, 11, density=0.02, kinds=[kind]), clean, placement)
    _compare(oracle, s, frame)


@pytest.mark.gpu
@pytest.mark.parametrize("reversed_z", [True, False])
def test_hostile_depth_conventional_and_reversed_z(oracle, reversed_z):
    w, h = 160, 90
    cam = synth.Camera(near=0.3, far=50.0, fov_y_deg=40.0, reversed_z=reversed_z)
    s = H.settings(oracle, w, h, cam=cam, thickness_modifier=3.0, intensity=1.7)
    depth = hostile_frame(w, h, 21, cam=cam, density=0.02)
    if not reversed_z:
        depth = (np.float32(1.0) - depth).astype(np.float32)   # keep NaN / inf etc.; values > 1 appear as well
    _compare(oracle, s, depth)


@pytest.mark.gpu
@pytest.mark.parametrize("f16_rounding", [0, 1])
def test_hostile_f16_depth_codes(oracle, f16_rounding):
    """MEAO_DEPTH_F16 accepts inf / NaN / negative codes."""
    w, h = 144, 80
    s = H.settings(oracle, w, h, depth_format=oracle.DEPTH_F16, f16_rounding=f16_rounding)
    base = oracle.encode_depth(synth.make("S2", w, h, seed=9), oracle.DEPTH_F16).copy()
    rng = np.random.default_rng(4)
    codes = np.array([0x7c00, 0xfc00, 0x7e00, 0xfe01, 0x8000, 0xbc00, 0x4500, 0x0001, 0x8001, 0x7bff], np.uint16)
    for _ in range(60):
        y, x = rng.integers(0, h), rng.integers(0, w)
        base[y, x] = codes[rng.integers(0, len(codes))]
    want = oracle.run(base, s)
    ao = H.component(s, depth_format=oracle.DEPTH_F16)
    try:
        got = ao.render(base)
        ok, bad = _nan_aware_equal(got, want["result"])
        assert ok, int(bad.sum())
        for i in (1, 2, 5, 10, 13, 14, 16):
            ok, bad = _nan_aware_equal(ao.debug_buffer(i), want[H.NAMES[i]])
            assert ok, (H.NAMES[i], int(bad.sum()))
    finally:
        ao.close()


@pytest.mark.gpu
@pytest.mark.parametrize("far,near", [(1.0e6, 0.01), (3.0e7, 1.0), (5.0e4, 0.001)])
def test_camera_with_huge_far_over_near(oracle, far, near):
    """far/near above 2^24: Linear01 depths next to the near plane flush to 0 in the f16 mips
    (1 / 0 = inf in Render.compute:140); ADVICE r1."""
    w, h = 192, 108
    cam = synth.Camera(near=near, far=far, fov_y_deg=50.0)
    s = H.settings(oracle, w, h, cam=cam)
    lin = synth._radial_linear01(w, h) * 0.2
    lin[20:40, 30:90] = near / far * 1.0001          # on the near plane
    lin[60:70, 100:150] = near / far * 3.0
    lin[80:, :20] = 2.0 ** -25
    depth = synth.linear01_to_raw(np.maximum(lin, near / far), cam)
    _compare(oracle, s, depth)


@pytest.mark.gpu
def test_hostile_flag_is_per_frame_and_per_call(oracle):
    """Batch with one hostile and one clean frame; then the same context on clean frames only,
    then hostile again -- through the plain and the pipelined (prefetch) paths."""
    w, h = 200, 120
    s = H.settings(oracle, w, h)
    clean = [synth.make("S2", w, h, seed=100 + i) for i in range(3)]
    dirty = [hostile_frame(w, h, 200 + i) for i in range(2)]
    ao = H.component(s, max_batch=2, pipelined=True)
    try:
        for batch in ([dirty[0], clean[0]], [clean[1], clean[2]], [clean[0], dirty[1]], [dirty[0], dirty[1]]):
            outs = ao.render_batch(batch)
            for f, d in enumerate(batch):
                _compare(oracle, s, d, ao=ao, frame=f, got=outs[f])
    finally:
        ao.close()


@pytest.mark.gpu
def test_hostile_frames_through_the_pipelined_path(oracle):
    torch = pytest.importorskip("torch")
    w, h = 256, 144
    s = H.settings(oracle, w, h)
    seq = [[hostile_frame(w, h, 300), synth.make("S2", w, h, seed=301)],
           [synth.make("S2", w, h, seed=302), synth.make("S2", w, h, seed=303)],
           [synth.make("S2", w, h, seed=304), hostile_frame(w, h, 305)]]
    dev = torch.device("cuda", 0)
    dd = [[torch.from_numpy(f).to(dev) for f in b] for b in seq]
    out = [[torch.empty((h, w), dtype=torch.uint8, device=dev) for _ in b] for b in seq]
    ao = H.component(s, max_batch=2, pipelined=True)
    try:
        stream = torch.cuda.current_stream(dev).cuda_stream
        for k in range(len(seq)):
            if k + 1 < len(seq):
                ao.prefetch_device([t.data_ptr() for t in dd[k + 1]])
            ao.execute_device([t.data_ptr() for t in dd[k]], [t.data_ptr() for t in out[k]], stream)
        torch.cuda.synchronize(dev)
        for k in range(len(seq)):
            for f in range(2):
                want = oracle.run(seq[k][f], s, result_only=True)["result"]
                assert np.array_equal(out[k][f].cpu().numpy(), want), (k, f)
    finally:
        ao.close()
