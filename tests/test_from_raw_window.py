"""The from-raw LoResDB window of the full-resolution upsample (round 6, meao_dev_upsample.hpp `ups_tile_from_raw`).

`LowDepth1[X, Y]` is `Linearize(Depth[2X, 2Y])` before the f16 store (Downsample1.compute:37-48, 64-70), and `Upsample.main` reads it as
`LoResDB` (Upsample.compute:54-72, AmbientOcclusion.cs:531).  A tile that lies inside the frame takes the 32 x 32 interior of that
window from its own hi-res operands and linearizes the 3-texel apron from the raw texels under it -- it never reads the buffer.  The
suites that existed before this path run through it wherever a frame has such tiles (>= 256 texels wide); the cases here are the
ones its special structure invites: hostile raw depth that only the full-resolution pass sees (odd texels: no frame flag, the
lane-redo path) inside and around from-raw tiles, far-plane texels in aprons and interiors under both Z conventions, frames whose
last tile row is partial (those tiles keep reading the buffer) next to frames that tile exactly, both tile heights of the pass,
every depth format, and the pipelined step whose last kernel carries the next downsample pass."""
import numpy as np
import pytest

from miniengineao_amd import _lib, synth
from tests import helpers as H

pytestmark = pytest.mark.gpu

# 64 x 64 tiles whatever the tile count (the default takes 64 x 32 tiles for calls this small)
TALL = {_lib.DEBUG_FINAL_SMALL_MAX_TILES: 0}


def _check(O, s, depth, debug=None):
    want = O.run(depth, s)
    ao = H.component(s, debug=debug)
    try:
        got = ao.render(depth)
        ok, bad = H.nan_aware_equal(got, want["result"])
        assert ok, f"result: {int(bad.sum())} texels differ, first {tuple(np.argwhere(bad)[0])}"
        for i in (1, 2):        # LinearDepth (built on demand) and LowDepth1 (still written: render and L2 -> L1 read it)
            ok, bad = H.nan_aware_equal(ao.debug_buffer(i), want[H.NAMES[i]])
            assert ok, (H.NAMES[i], int(bad.sum()))
    finally:
        ao.close()


def _odd_texels_only(frame, clean):
    yy, xx = np.mgrid[0:frame.shape[0], 0:frame.shape[1]]
    level = ((yy & 1) == 0) & ((xx & 1) == 0)
    return np.where(~level, frame, clean).astype(np.float32)


@pytest.mark.parametrize("debug", [None, TALL], ids=["tiles64x32", "tiles64x64"])
@pytest.mark.parametrize("kind", ["nan", "pinf", "ninf", "neg", "big", "huge", "nhuge", "denorm", "negzero", "zero_den", "tiny_den"])
def test_hostile_odd_texels_around_from_raw_tiles(oracle, kind, debug):
    """The levels stay clean (no frame flag): the exact-division instance runs, its from-raw windows next to lanes that redo."""
    w, h = 320, 192
    s = H.settings(oracle, w, h)
    clean = synth.make("S2", w, h, seed=31)
    frame = _odd_texels_only(H.hostile_frame(w, h, 31, density=0.02, kinds=[kind]), clean)
    _check(oracle, s, frame, debug)


@pytest.mark.parametrize("debug", [None, TALL], ids=["tiles64x32", "tiles64x64"])
@pytest.mark.parametrize("ao_format,f16_rounding", [(0, 0), (1, 0), (0, 1)])
def test_hostile_level_texels_take_the_ieee_instance_of_the_from_raw_tile(oracle, ao_format, f16_rounding, debug):
    w, h = 320, 192
    s = H.settings(oracle, w, h, ao_format=ao_format, f16_rounding=f16_rounding)
    _check(oracle, s, H.hostile_frame(w, h, 32, density=0.01), debug)


@pytest.mark.parametrize("debug", [None, TALL], ids=["tiles64x32", "tiles64x64"])
@pytest.mark.parametrize("reversed_z", [True, False])
def test_far_plane_texels_in_aprons_and_interiors(oracle, reversed_z, debug):
    """Sky texels (DS1:41-45) as stripes that cross tile borders: apron texels, interior texels, odd and even positions."""
    w, h = 384, 256
    cam = synth.Camera(near=0.3, far=80.0, fov_y_deg=50.0, reversed_z=reversed_z)
    s = H.settings(oracle, w, h, cam=cam)
    depth = synth.occluder_field(w, h, seed=33, cam=cam).copy()
    sky = np.float32(0.0 if reversed_z else 1.0)
    depth[:, 60:70] = sky           # across the x = 64 tile border
    depth[:, 122:129] = sky
    depth[58:72, :] = sky           # across the y = 64 tile border
    depth[127:130, 200:330] = sky
    depth[::17, ::13] = sky         # isolated ones
    _check(oracle, s, depth, debug)


@pytest.mark.parametrize("w,h", [(320, 192), (320, 200), (320, 130), (448, 64), (260, 192), (512, 320)])
@pytest.mark.parametrize("debug", [None, TALL], ids=["tiles64x32", "tiles64x64"])
def test_exact_and_partial_last_tile_rows(oracle, w, h, debug):
    """Frames that tile exactly (every interior-column tile is from-raw) next to frames whose last tile row is partial (those tiles
    read LowDepth1) and frames whose bottom apron clamps to the last LowDepth1 row."""
    s = H.settings(oracle, w, h)
    _check(oracle, s, synth.make("S2", w, h, seed=w + h), debug)


@pytest.mark.parametrize("fmt", ["unorm16", "unorm24", "f16"])
def test_depth_formats_through_tall_from_raw_tiles(oracle, fmt):
    from tests.test_depth_formats import FORMATS
    w, h = 320, 192
    raw = synth.occluder_field(w, h, seed=35).copy()
    raw[100:140, 150:] = 0.0                                          # sky (reversed Z), across tile borders
    depth = oracle.encode_depth(raw, FORMATS[fmt])
    s = H.settings(oracle, w, h, depth_format=FORMATS[fmt])
    want = oracle.run(depth, s)
    ao = H.component(s, debug=TALL, depth_format=FORMATS[fmt])
    try:
        assert np.array_equal(ao.render(depth), want["result"])
        for i in (1, 2):
            assert np.array_equal(ao.debug_buffer(i), want[H.NAMES[i]]), H.NAMES[i]
    finally:
        ao.close()


def test_pipelined_batches_through_the_fused_last_kernel(oracle):
    """Three batches of four frames, each announced to the call before it: the last kernel of a call evaluates from-raw tiles of ITS
    frames while it writes the levels of the next ones; one frame carries hostile odd texels, one hostile level texels."""
    torch = pytest.importorskip("torch")
    w, h = 384, 256             # 24 lean downsample tiles for 24 upsample tiles: the fused form applies (fused_downsample_applicable)
    s = H.settings(oracle, w, h)
    clean = synth.make("S2", w, h, seed=40)
    seq = [[synth.make("S2", w, h, seed=41 + 4 * k + f) for f in range(4)] for k in range(3)]
    seq[1][2] = _odd_texels_only(H.hostile_frame(w, h, 60, density=0.02), clean)
    seq[2][1] = H.hostile_frame(w, h, 61)
    dev = torch.device("cuda", 0)
    dd = [[torch.from_numpy(f).to(dev) for f in b] for b in seq]
    out = [[torch.empty((h, w), dtype=torch.uint8, device=dev) for _ in b] for b in seq]
    ao = H.component(s, max_batch=4, pipelined=True)
    try:
        stream = torch.cuda.current_stream(dev).cuda_stream
        for k in range(len(seq)):
            if k + 1 < len(seq):
                ao.prefetch_device([t.data_ptr() for t in dd[k + 1]])
            ao.execute_device([t.data_ptr() for t in dd[k]], [t.data_ptr() for t in out[k]], stream)
        torch.cuda.synchronize(dev)
        for k in range(len(seq)):
            for f in range(4):
                want = oracle.run(seq[k][f], s, result_only=True)["result"]
                ok, bad = H.nan_aware_equal(out[k][f].cpu().numpy(), want)
                assert ok, (k, f, int(bad.sum()))
    finally:
        ao.close()
