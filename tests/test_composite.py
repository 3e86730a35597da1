"""Composite step: the raster blits that consume the AO texture
(Blit.shader passes 1-3, PushCompositeCommands AO.cs:822-839) as one streaming kernel."""
import numpy as np
import pytest

from miniengineao_amd import synth
from tests import helpers as H


def _targets(rng, h, w):
    color = (rng.random((h, w, 4)) * 8.0).astype(np.float16).view(np.uint16)
    color[0, 0] = [0x7c00, 0xfc00, 0x0001, 0x8000]           # inf, -inf, smallest subnormal, -0
    gbuf = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
    return color, gbuf


def test_oracle_composite_semantics(oracle):
    rng = np.random.default_rng(0)
    h, w = 9, 13
    ao = rng.integers(0, 256, (h, w), dtype=np.uint8)
    ao[0, :3] = [0, 255, 128]
    color, gbuf = _targets(rng, h, w)
    c0 = color.copy()
    oracle.composite(ao, c0, 0)
    a32 = ao.astype(np.float32) / np.float32(255)
    with np.errstate(invalid="ignore"):                      # the probe texel multiplies inf by 0
        want = (color.view(np.float16).astype(np.float32) * a32[..., None]).astype(np.float16).view(np.uint16)
    assert np.array_equal(c0[1:], want[1:])                  # numpy f32->f16 is RTNE
    assert (c0[0, 1] == color[0, 1]).all()                   # ao = 255/255 = 1 keeps the texel
    c2 = color.copy()
    oracle.composite(ao, c2, 2)
    assert (c2[0, 1] == 0x3c00).all() and (c2[0, 0] == 0).all()
    c1, g1 = color.copy(), gbuf.copy()
    oracle.composite(ao, c1, 1, gbuffer0_rgba8=g1)
    assert np.array_equal(c1[..., 3], color[..., 3]) and np.array_equal(g1[..., :3], gbuf[..., :3])
    keep = np.float32(1) - (np.float32(1) - a32)
    want_a = np.floor(np.clip(gbuf[..., 3].astype(np.float32) / np.float32(255) * keep, 0, 1) * np.float32(255) + np.float32(0.5))
    assert np.array_equal(g1[..., 3], want_a.astype(np.uint8))


@pytest.mark.gpu
@pytest.mark.parametrize("ao_format", [0, 1])
@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("w,h", [(64, 48), (131, 77)])
def test_gpu_composite_matches_oracle(oracle, ao_format, mode, w, h):
    rng = np.random.default_rng(mode * 10 + ao_format)
    s = H.settings(oracle, w, h, ao_format=ao_format, intensity=1.5)
    depth = synth.make("S2", w, h, seed=3)
    ao_comp = H.component(s)
    try:
        ao = ao_comp.render(depth)
        color, gbuf = _targets(rng, h, w)
        want_c, want_g = color.copy(), gbuf.copy()
        oracle.composite(ao, want_c, mode, ao_format, want_g if mode == 1 else None)
        got_c, got_g = color.copy(), gbuf.copy()
        ao_comp.ambientOnly = mode == 1
        ao_comp.composite(ao, got_c, got_g if mode == 1 else None, debug=(mode == 2))
        nan = lambda x: ((x & 0x7fff) > 0x7c00)             # noqa: E731  NaN payloads are not compared
        assert np.array_equal(np.where(nan(got_c), 0x7e00, got_c), np.where(nan(want_c), 0x7e00, want_c))
        assert np.array_equal(got_g, want_g)
    finally:
        ao_comp.close()


# ---- the pipelined composite: rides inside the next execute's render kernel ------------------------

@pytest.mark.gpu
@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("ao_format", [0, 1])
def test_enqueued_composite_rides_in_the_next_render_and_matches_oracle(oracle, mode, ao_format):
    torch = pytest.importorskip("torch")
    from tests import helpers as H
    from miniengineao_amd import synth
    w, h, n = 200, 88, 3
    s = H.settings(oracle, w, h, ao_format=ao_format)
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(31 + mode)
    frames = [synth.make("S2", w, h, seed=900 + f) for f in range(n)]
    next_frames = [synth.make("S2", w, h, seed=950 + f) for f in range(n)]
    want_ao = [oracle.run(f, s, result_only=True)["result"] for f in frames]
    want_next = [oracle.run(f, s, result_only=True)["result"] for f in next_frames]
    colors = [(rng.random((h, w, 4)) * 3.0).astype(np.float16).view(np.uint16) for _ in range(n)]
    gbufs = [rng.integers(0, 256, (h, w, 4), dtype=np.uint8) for _ in range(n)]
    want_c, want_g = [c.copy() for c in colors], [g.copy() for g in gbufs]
    for f in range(n):
        oracle.composite(want_ao[f], want_c[f], mode, ao_format, want_g[f] if mode == 1 else None)
    ao_dt = torch.uint8 if ao_format == 0 else torch.int16
    dd = [torch.from_numpy(f).to(dev) for f in frames]
    dn = [torch.from_numpy(f).to(dev) for f in next_frames]
    out = [torch.zeros((h, w), dtype=ao_dt, device=dev) for _ in range(n)]
    out2 = [torch.zeros((h, w), dtype=ao_dt, device=dev) for _ in range(n)]
    dc = [torch.from_numpy(c.view(np.int16)).to(dev) for c in colors]
    dg = [torch.from_numpy(g).to(dev) for g in gbufs]
    st = torch.cuda.current_stream(dev).cuda_stream
    ao = H.component(s, max_batch=n)
    try:
        ao.execute_device([t.data_ptr() for t in dd], [t.data_ptr() for t in out], st)
        ao.composite_enqueue_device(mode, [t.data_ptr() for t in out], [t.data_ptr() for t in dc],
                                    [t.data_ptr() for t in dg] if mode == 1 else None)
        ao.execute_device([t.data_ptr() for t in dn], [t.data_ptr() for t in out2], st)   # carries the composite
        torch.cuda.synchronize(dev)
        for f in range(n):
            assert np.array_equal(out2[f].cpu().numpy().view(want_next[f].dtype), want_next[f]), f
            assert np.array_equal(dc[f].cpu().numpy().view(np.uint16), want_c[f]), (f, "color")
            assert np.array_equal(dg[f].cpu().numpy(), want_g[f]), (f, "gbuffer0")
    finally:
        ao.close()


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,n_composite,n_render", [(131, 77, 2, 2), (640, 360, 1, 1), (200, 88, 2, 3), (96, 64, 3, 1)])
def test_carried_composite_shapes(oracle, w, h, n_composite, n_render):
    """Multiply mode inside the render texel loop: an odd pixel count (131 x 77: the half pair at the end),
    more pairs per lane than the loop takes (640 x 360: the remainder runs before the tile), and a
    carrying call with a different frame count than the composite batch (falls back to "composite first")."""
    torch = pytest.importorskip("torch")
    from tests import helpers as H
    from miniengineao_amd import synth
    mode = 0          # MEAO_COMPOSITE_MULTIPLY
    s = H.settings(oracle, w, h)
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(w + h)
    nmax = max(n_composite, n_render)
    frames = [synth.make("S2", w, h, seed=300 + f) for f in range(nmax)]
    want_ao = [oracle.run(f, s, result_only=True)["result"] for f in frames]
    colors = [(rng.random((h, w, 4)) * 3.0).astype(np.float16).view(np.uint16) for _ in range(n_composite)]
    want_c = [c.copy() for c in colors]
    for f in range(n_composite):
        oracle.composite(want_ao[f], want_c[f], mode, 0, None)
    dd = [torch.from_numpy(f).to(dev) for f in frames]
    out = [torch.zeros((h, w), dtype=torch.uint8, device=dev) for _ in range(nmax)]
    out2 = [torch.zeros((h, w), dtype=torch.uint8, device=dev) for _ in range(nmax)]
    dc = [torch.from_numpy(c.view(np.int16)).to(dev) for c in colors]
    st = torch.cuda.current_stream(dev).cuda_stream
    ao = H.component(s, max_batch=nmax)
    try:
        ao.execute_device([t.data_ptr() for t in dd[:n_composite]], [t.data_ptr() for t in out[:n_composite]], st)
        ao.composite_enqueue_device(mode, [t.data_ptr() for t in out[:n_composite]], [t.data_ptr() for t in dc], None)
        ao.execute_device([t.data_ptr() for t in dd[:n_render]], [t.data_ptr() for t in out2[:n_render]], st)
        torch.cuda.synchronize(dev)
        for f in range(n_render):
            assert np.array_equal(out2[f].cpu().numpy(), want_ao[f]), f
        for f in range(n_composite):
            assert np.array_equal(dc[f].cpu().numpy().view(np.uint16), want_c[f]), (f, "color")
    finally:
        ao.close()


@pytest.mark.gpu
def test_enqueued_composite_flush_rules(oracle):
    """An explicit flush and a second enqueue run a waiting batch; meao_destroy discards one (its targets are
    caller memory that is usually gone by then -- include/meao.h)."""
    torch = pytest.importorskip("torch")
    from tests import helpers as H
    from miniengineao_amd import synth
    w, h = 96, 64
    s = H.settings(oracle, w, h)
    dev = torch.device("cuda", 0)
    depth = synth.make("S1", w, h)
    want_ao = oracle.run(depth, s, result_only=True)["result"]
    base = (np.random.default_rng(5).random((h, w, 4)) * 2.0).astype(np.float16).view(np.uint16)
    want = base.copy()
    oracle.composite(want_ao, want, 0)
    d = torch.from_numpy(depth).to(dev)
    out = torch.zeros((h, w), dtype=torch.uint8, device=dev)
    cols = [torch.from_numpy(base.view(np.int16).copy()).to(dev) for _ in range(3)]
    ao = H.component(s)
    ao.execute_device([d.data_ptr()], [out.data_ptr()])
    ao.composite_enqueue_device(0, [out.data_ptr()], [cols[0].data_ptr()])
    ao.composite_flush()                                             # 1: explicit flush
    ao.composite_enqueue_device(0, [out.data_ptr()], [cols[1].data_ptr()])
    ao.composite_enqueue_device(0, [out.data_ptr()], [cols[2].data_ptr()])   # 2: pushes the older one out
    ao.close()                                                       # 3: destroy discards what still waits
    torch.cuda.synchronize(dev)
    for k in range(2):
        assert np.array_equal(cols[k].cpu().numpy().view(np.uint16), want), k
    assert np.array_equal(cols[2].cpu().numpy().view(np.uint16), base), "a waiting batch must not be written at destroy"
