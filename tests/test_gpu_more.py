"""More GPU parity: randomized configurations, the device-pointer / stream path used by
bench.py, context independence, error behaviour, the compiled C++ host, full-size 8K fp16."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from miniengineao_amd import _lib as L
from miniengineao_amd import synth
from tests import helpers as H

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", range(40))
def test_randomized_configurations(oracle, seed):
    """Random size (incl. widths that are not multiples of 4: scalar load/store paths), camera,
    component properties inside the reference's Range attributes (AO.cs:20-52), storage modes,
    level count, with sky patches: every buffer bit-exact."""
    rng = np.random.default_rng(1000 + seed)
    w, h = int(rng.integers(1, 420)), int(rng.integers(1, 300))
    if seed % 5 == 0:
        w = int(rng.integers(1, 90)) * 4                      # vector paths on small images too
    reversed_z = bool(rng.integers(0, 2))
    cam = synth.Camera(near=float(rng.uniform(0.01, 0.6)), far=float(rng.uniform(10, 2000)),
                       fov_y_deg=float(rng.uniform(12, 90)), reversed_z=reversed_z)
    s = H.settings(oracle, w, h, cam=cam, ao_format=int(rng.integers(0, 2)), f16_rounding=int(rng.integers(0, 2)),
                   num_levels=int(rng.integers(1, 5)), noise_filter_tolerance=float(rng.uniform(-8, 0)),
                   blur_tolerance=float(rng.uniform(-8, -1)), upsample_tolerance=float(rng.uniform(-12, -1)),
                   thickness_modifier=float(rng.uniform(1, 10)), intensity=float(rng.uniform(0, 2)))
    if seed % 2:                                               # the variants of SURVEY 8f #4
        s.hq_levels = int(rng.integers(0, s.num_levels + 1))
        s.sample_set = int(rng.integers(0, 2))
        s.single_pass_stereo = bool(rng.integers(0, 2))
    depth = synth.occluder_field(w, h, seed=seed, n_rects=24, n_discs=24, cam=cam)
    if seed % 3 == 0:                                          # sky (1e5; overflows f16)
        x0, y0 = int(rng.integers(0, w)), int(rng.integers(0, h))
        depth[y0:y0 + 40, x0:x0 + 60] = 0.0 if reversed_z else 1.0
    want = oracle.run(depth, s)
    ao = H.component(s)
    try:
        got = ao.render(depth)
        assert np.array_equal(got, want["result"]), (seed, w, h, H.diff_report("result", got, want["result"]))
        for i in H.valid_debug_ids(s.num_levels, s.hq_levels):
            g = ao.debug_buffer(i)
            assert np.array_equal(g, want[H.NAMES[i]]), (seed, w, h, H.diff_report(H.NAMES[i], g, want[H.NAMES[i]]))
    finally:
        ao.close()


def test_device_pointers_and_streams_with_torch(oracle):
    """The path bench.py uses: torch owns device memory and the stream, the library gets raw
    addresses; batched, asynchronous, results identical to the host-pointer path."""
    import torch
    w, h, n = 512, 288, 6
    s = H.settings(oracle, w, h)
    dev = torch.device("cuda", 0)
    depths = [synth.make("S2", w, h, seed=50 + f) for f in range(n)]
    d_dev = [torch.from_numpy(d).to(dev) for d in depths]
    o_dev = [torch.zeros((h, w), dtype=torch.uint8, device=dev) for _ in range(n)]
    side = torch.cuda.Stream(dev)
    ao = H.component(s, max_batch=8)
    try:
        torch.cuda.synchronize(dev)
        ao.execute_device([t.data_ptr() for t in d_dev], [t.data_ptr() for t in o_dev], side.cuda_stream)
        ao.synchronize(side.cuda_stream)
        for f in range(n):
            want = oracle.run(depths[f], s, result_only=True)["result"]
            assert np.array_equal(o_dev[f].cpu().numpy(), want), f
        assert np.array_equal(ao.debug_buffer(17, frame=3), o_dev[3].cpu().numpy())
    finally:
        ao.close()


def test_contexts_are_independent(oracle):
    """Two differently sized contexts used alternately (the reference's shared statics,
    AO.cs:136-137,592-593, make this unsafe there)."""
    s1, s2 = H.settings(oracle, 200, 120, intensity=0.7), H.settings(oracle, 131, 257, thickness_modifier=4.0)
    d1, d2 = synth.make("S2", 200, 120, seed=1), synth.make("S2", 131, 257, seed=2)
    a1, a2 = H.component(s1), H.component(s2)
    try:
        w1, w2 = oracle.run(d1, s1)["result"], oracle.run(d2, s2)["result"]
        for _ in range(3):
            assert np.array_equal(a1.render(d1), w1)
            assert np.array_equal(a2.render(d2), w2)
    finally:
        a1.close()
        a2.close()


def test_error_behaviour_on_device(oracle):
    s = H.settings(oracle, 64, 48, num_levels=2)
    ao = H.component(s, max_batch=2)
    lib = L.load()
    try:
        depth = synth.make("S1", 64, 48)
        with pytest.raises(L.MeaoError) as e:                  # nothing executed yet
            ao.debug_buffer(2)
        assert e.value.status == L.ERR_INVALID_ARGUMENT
        ao.render(depth)
        d = L.Desc()
        small = np.zeros(10, np.uint8)
        rc = lib.meao_get_intermediate(ao._ctx, 0, 1, small.ctypes.data, small.nbytes, L.MEM_HOST, C.byref(d))
        assert rc == L.ERR_BUFFER_TOO_SMALL and d.bytes == 64 * 48 * 2
        with pytest.raises(L.MeaoError) as e:                  # level 3 not rendered with num_levels=2
            ao.debug_buffer(12)
        assert e.value.status == L.ERR_UNSUPPORTED
        with pytest.raises(L.MeaoError):
            ao.debug_buffer(2, frame=1)                        # only one frame was executed
        with pytest.raises(L.MeaoError) as e:
            ao.render_batch([depth, depth, depth])             # n > max_batch
        assert e.value.status == L.ERR_INVALID_ARGUMENT
        with pytest.raises(ValueError):
            ao.render(np.zeros((10, 10), np.float32))
        ao.intensity = float("nan")
        with pytest.raises(L.MeaoError):
            ao.render(depth)
        assert b"non-finite" in lib.meao_last_error(ao._ctx)
    finally:
        ao.close()


def test_compiled_cpp_host_program(oracle, tmp_path):
    """examples/ao_host_demo.cpp (g++, include/meao.hpp, no Python in the loop) vs the oracle."""
    from miniengineao_amd import build
    exe = build.build_host_demo()
    w, h = 333, 211
    depth = synth.make("S2", w, h, seed=77)
    (tmp_path / "d.f32").write_bytes(depth.tobytes())
    out = tmp_path / "ao.u8"
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.dirname(exe) + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    proc = subprocess.run([exe, str(w), str(h), str(tmp_path / "d.f32"), str(out), "1.25", "2.5"],
                          capture_output=True, text=True, env=env)
    assert proc.returncode == 0, proc.stderr
    s = H.settings(oracle, w, h, intensity=1.25, thickness_modifier=2.5)
    want = oracle.run(depth, s, result_only=True)["result"]
    got = np.frombuffer(out.read_bytes(), np.uint8).reshape(h, w)
    assert np.array_equal(got, want), H.diff_report("result", got, want)


def test_8k_fp16_full_frame(oracle):
    """BASELINE config 5: 7680x4320, fp16 AO storage, full multi-scale, bit-exact."""
    w, h = 7680, 4320
    depth = synth.make("S2", w, h)
    s = H.settings(oracle, w, h, ao_format=1)
    want = oracle.run(depth, s, nthreads=oracle.host_cores(), result_only=True)["result"]
    ao = H.component(s)
    try:
        got = ao.render(depth)
    finally:
        ao.close()
    assert np.array_equal(got, want), H.diff_report("result", got, want)


@pytest.mark.parametrize("variant", [dict(), dict(ao_format=1, f16_rounding=1), dict(hq_levels=1, num_levels=3)])
@pytest.mark.parametrize("w,h,batch", [(322, 182, 1), (256, 128, 3), (131, 77, 2)])
def test_pipelined_downsample(oracle, variant, w, h, batch):
    """meao_prefetch_batch: the downsample pass of the next call runs inside the last upsample kernel
    of the current one.  A stream of different frames, a mispredicted announcement, a property change
    and a resize in between: every result and every intermediate stays bit-exact."""
    import torch
    dev = torch.device("cuda", 0)
    s = H.settings(oracle, w, h, **variant)
    ao = H.component(s, max_batch=batch)
    dt = torch.uint8 if s.ao_format == 0 else torch.int16
    try:
        sets = [[synth.make("S2", w, h, seed=100 * k + f) for f in range(batch)] for k in range(5)]
        d_in = [[torch.from_numpy(f).to(dev) for f in fs] for fs in sets]
        d_out = [torch.zeros((h, w), dtype=dt, device=dev) for _ in range(batch)]
        stream = torch.cuda.current_stream(dev).cuda_stream
        ptrs = lambda k: [t.data_ptr() for t in d_in[k]]                      # noqa: E731
        optr = [t.data_ptr() for t in d_out]

        def run(k, announce=None, settings=s):
            if announce is not None:
                ao.prefetch_device(ptrs(announce))
            ao.execute_device(ptrs(k), optr, stream)
            torch.cuda.synchronize(dev)
            for f in range(batch):
                want = oracle.run(sets[k][f], settings, result_only=True)["result"]
                got = d_out[f].cpu().numpy().view(want.dtype)
                assert np.array_equal(got, want), (k, f, H.diff_report("result", got, want))

        run(0, announce=1)                 # downsample of set 1 rides in this call
        run(1, announce=2)                 # consumes it, carries set 2
        want = oracle.run(sets[1][batch - 1], s)
        for i in H.valid_debug_ids(s.num_levels, s.hq_levels):               # intermediates come from the prefetched set
            assert np.array_equal(ao.debug_buffer(i, frame=batch - 1), want[H.NAMES[i]]), i
        run(2, announce=4)                 # consumes set 2, carries set 4 ...
        run(3)                             # ... but set 3 arrives: mispredicted, runs its own downsample
        run(4)                             # the stale prefetch was dropped by the previous call
        run(0, announce=0)                 # steady state of bench.py: the same frames announced again
        run(0, announce=0)
        ao.intensity = 0.7                 # property change between announce-carrying call and consumer
        s2 = H.settings(oracle, w, h, intensity=0.7, **variant)
        run(0, settings=s2)
        run(1, announce=2, settings=s2)
        run(2, settings=s2)
        # a resize between the announcing call and the consumer drops the prefetched set as well
        ao.prefetch_device(ptrs(3))
        ao.execute_device(ptrs(2), optr, stream)
        torch.cuda.synchronize(dev)
        w2, h2 = w - 6, h - 3
        ao.resize(w2, h2)
        s3 = H.settings(oracle, w2, h2, intensity=0.7, **variant)
        ao.projection00 = s3.proj00
        small = [synth.make("S2", w2, h2, seed=900 + f) for f in range(batch)]
        outs = ao.render_batch(small)
        for f in range(batch):
            assert np.array_equal(outs[f], oracle.run(small[f], s3, result_only=True)["result"]), f
    finally:
        ao.close()


@pytest.mark.parametrize("depth_format", [0, 3, 1])             # f32, f16, unorm16
@pytest.mark.parametrize("w,h,batch", [(132, 40, 2), (260, 36, 1), (512, 256, 2), (192, 108, 3), (128, 96, 2)])
def test_pipelined_downsample_tile_counts_and_formats(oracle, w, h, batch, depth_format):
    """The last kernel carries the next call's downsample tiles one per workgroup with their loads issued
    inside the upsample tile (16-byte f32 rows), and falls back to "tiles first" otherwise: more downsample
    tiles than upsample tiles (132x40: 4 vs 3; 260x36: 6 vs 5), fewer (128x96: 3 vs 4: one workgroup carries
    nothing), 16-bit depth formats, odd batches."""
    import torch
    dev = torch.device("cuda", 0)
    s = H.settings(oracle, w, h, depth_format=depth_format)
    ao = H.component(s, max_batch=batch, depth_format=depth_format, pipelined=True)
    try:
        sets = [[oracle.encode_depth(synth.make("S2", w, h, seed=31 * k + f), depth_format) for f in range(batch)]
                for k in range(4)]
        as_torch = lambda f: torch.from_numpy(f.view(np.int16) if f.dtype == np.uint16 else f).to(dev)   # noqa: E731
        d_in = [[as_torch(np.ascontiguousarray(f)) for f in fs] for fs in sets]
        d_out = [torch.zeros((h, w), dtype=torch.uint8, device=dev) for _ in range(batch)]
        stream = torch.cuda.current_stream(dev).cuda_stream
        for k in range(4):
            if k + 1 < 4:
                ao.prefetch_device([t.data_ptr() for t in d_in[k + 1]])
            ao.execute_device([t.data_ptr() for t in d_in[k]], [t.data_ptr() for t in d_out], stream)
            torch.cuda.synchronize(dev)
            for f in range(batch):
                want = oracle.run(sets[k][f], s, result_only=(k != 2))
                assert np.array_equal(d_out[f].cpu().numpy(), want["result"]), (k, f)
                if k == 2:          # every intermediate of a consumed, carried downsample set
                    for i in H.valid_debug_ids(s.num_levels, s.hq_levels):
                        assert np.array_equal(ao.debug_buffer(i, frame=f), want[H.NAMES[i]]), (H.NAMES[i], f)
    finally:
        ao.close()


@pytest.mark.parametrize("mode", ["separate", "two_level", "three_level"])
@pytest.mark.parametrize("w,h,batch", [(203, 117, 3), (256, 128, 1), (640, 360, 2), (67, 45, 2)])
def test_every_blend_launch_structure_is_bit_exact(oracle, mode, w, h, batch):
    """The three blend passes run as three launches, as L4->L3 inside L3->L2, or all inside the L2->L1
    launch (chosen by the size of the call; forced here through meao_debug_set).  Every
    buffer of every frame -- one of them hostile -- must equal the oracle's in each structure."""
    if mode == "separate":
        debug = {L.DEBUG_FUSE_COARSE_BLEND: 0}
    else:
        debug = {L.DEBUG_NESTED_MAX_TILES: 0 if mode == "two_level" else 1000000}
    s = H.settings(oracle, w, h)
    frames = [synth.make("S2", w, h, seed=50 + f) for f in range(batch)]
    frames[-1] = H.hostile_frame(w, h, 77, density=0.01)
    ao = H.component(s, max_batch=batch, debug=debug)
    try:
        outs = ao.render_batch(frames)
        for f in range(batch):
            want = oracle.run(frames[f], s)
            ok, bad = H.nan_aware_equal(outs[f], want["result"])
            assert ok, (f, int(bad.sum()))
            for i in H.valid_debug_ids(s.num_levels, s.hq_levels):
                ok, bad = H.nan_aware_equal(ao.debug_buffer(i, frame=f), want[H.NAMES[i]])
                assert ok, (H.NAMES[i], f, int(bad.sum()))
    finally:
        ao.close()


@pytest.mark.parametrize("small_tiles", [0, 1000000])
@pytest.mark.parametrize("variant", [dict(), dict(ao_format=1, f16_rounding=1), dict(num_levels=2)])
@pytest.mark.parametrize("w,h,batch", [(203, 117, 2), (512, 300, 1), (131, 77, 3)])
def test_both_render_tilings_are_bit_exact(oracle, small_tiles, variant, w, h, batch):
    """Calls with few tiles use 128 x 8 render tiles (render_small_kernel) and 64 x 32 tiles in the final
    upsample pass (upsample_final_small_kernel), larger ones 128 x 32 and 64 x 64; the thresholds are forced
    either way here (meao_debug_set)."""
    debug = {L.DEBUG_RENDER_SMALL_MAX_TILES: small_tiles,
             L.DEBUG_FINAL_SMALL_MAX_TILES: small_tiles,     # final pass: 64 x 32 / 64 x 64 tiles
             L.DEBUG_DS_SMALL_MAX_TILES: small_tiles}        # downsample pass: 128 x 8 / 128 x 32 tiles
    s = H.settings(oracle, w, h, **variant)
    frames = [synth.make("S2", w, h, seed=70 + f) for f in range(batch)]
    frames[0] = H.hostile_frame(w, h, 78, density=0.01)
    ao = H.component(s, max_batch=batch, debug=debug)
    try:
        outs = ao.render_batch(frames)
        for f in range(batch):
            want = oracle.run(frames[f], s)
            ok, bad = H.nan_aware_equal(outs[f], want["result"])
            assert ok, (f, int(bad.sum()))
            for i in H.valid_debug_ids(s.num_levels, s.hq_levels):
                ok, bad = H.nan_aware_equal(ao.debug_buffer(i, frame=f), want[H.NAMES[i]])
                assert ok, (H.NAMES[i], f, int(bad.sum()))
    finally:
        ao.close()


@pytest.mark.parametrize("variant", [dict(), dict(ao_format=1, f16_rounding=1), dict(num_levels=2), dict(hq_levels=2)])
@pytest.mark.parametrize("w,h,batch", [(644, 364, 3), (1280, 720, 2), (203, 117, 2), (2048, 1152, 1)])
def test_l2_to_l1_blend_with_64x64_tiles_is_bit_exact(oracle, variant, w, h, batch):
    """MEAO_DEBUG_BLEND_TALL_MIN_TILES: the L2 -> L1 pass with the full-resolution pass's 64 x 64 tiles (upsample_blend_tall_kernel),
    plain and pipelined, a hostile frame, every buffer of every frame against the oracle."""
    import torch
    dev = torch.device("cuda", 0)
    s = H.settings(oracle, w, h, **variant)
    frames = [synth.make("S2", w, h, seed=40 + f) for f in range(batch)]
    frames[-1] = H.hostile_frame(w, h, 43, density=0.004)
    elem = torch.uint8 if s.ao_format == oracle.AO_R8 else torch.int16
    for pipelined in (False, True):
        ao = H.component(s, max_batch=batch, pipelined=pipelined, debug={L.DEBUG_BLEND_TALL_MIN_TILES: 1, L.DEBUG_NESTED_MAX_TILES: 0})
        try:
            dd = [torch.from_numpy(f).to(dev) for f in frames]
            out = [torch.zeros((h, w), dtype=elem, device=dev) for _ in frames]
            st = torch.cuda.Stream(dev)
            for k in range(2):
                if pipelined:
                    ao.prefetch_device([t.data_ptr() for t in dd])
                ao.execute_device([t.data_ptr() for t in dd], [t.data_ptr() for t in out], st.cuda_stream)
            st.synchronize()
            for f in range(batch):
                want = oracle.run(frames[f], s)
                ok, bad = H.nan_aware_equal(out[f].cpu().numpy().view(want["result"].dtype), want["result"])
                assert ok, (pipelined, f, int(bad.sum()))
                for i in H.valid_debug_ids(s.num_levels, s.hq_levels):
                    ok, bad = H.nan_aware_equal(ao.debug_buffer(i, frame=f), want[H.NAMES[i]])
                    assert ok, (H.NAMES[i], pipelined, f, int(bad.sum()))
        finally:
            ao.close()


def test_sampled_profiling_brackets_every_nth_execute(oracle):
    """meao_set_profiling(N > 1): events around the passes of executes 0, N, 2N, ... only; the others run bare."""
    w, h = 322, 182
    s = H.settings(oracle, w, h)
    depth = synth.make("S2", w, h, seed=4)
    want = oracle.run(depth, s, result_only=True)["result"]
    ao = H.component(s)
    try:
        for period, calls, sampled in ((3, 7, 3), (1, 4, 4), (4, 4, 1), (2, 5, 3)):
            ao.set_profiling(period)
            for _ in range(calls):
                assert np.array_equal(ao.render(depth), want)
            ms, n = ao.pass_times_ms()
            assert n == sampled and sum(ms) > 0, (period, n)
        ao.set_profiling(0)                 # off: the window of the last measurement stays as it was
        for _ in range(3):
            assert np.array_equal(ao.render(depth), want)
        assert ao.pass_times_ms()[1] == 3
    finally:
        ao.close()


def test_profiling_of_selected_passes_only(oracle):
    """MEAO_DEBUG_PROFILE_PASS_MASK: only the launch slots named by the mask are bracketed with events (a host that wants one
    kernel's duration in a throughput run pays for one pair of marker packets, not eight); 0 = all again.  Results unchanged."""
    w, h = 644, 364
    s = H.settings(oracle, w, h)
    depth = synth.make("S2", w, h, seed=5)
    want = oracle.run(depth, s, result_only=True)["result"]
    ao = H.component(s)
    try:
        names = list(L.PASS_NAMES)
        final, render = names.index("upsample_L1_to_L0"), names.index("render")
        for mask, expect in ((1 << final, {final}), ((1 << final) | (1 << render), {final, render}), (0, None)):
            ao.debug_set(L.DEBUG_PROFILE_PASS_MASK, mask)
            ao.set_profiling(True)
            for _ in range(3):
                assert np.array_equal(ao.render(depth), want)
            ms, n = ao.pass_times_ms()
            timed = {k for k in range(len(names)) if ms[k] > 0}
            assert n == 3 and (timed == expect if expect is not None else {final, render, names.index("downsample")} <= timed), (mask, ms)
    finally:
        ao.close()


# ---- round 2: contract enforcement, robustness of the boundary ------------------------------------

def test_prefetched_downsample_is_not_used_from_another_stream(oracle):
    """The pipelining contract is enforced: a consumer on a different stream than the carrying execute
    runs its own downsample pass (and is still correct); the same stream consumes the prefetched set."""
    import torch
    dev = torch.device("cuda", 0)
    w, h = 256, 128
    s = H.settings(oracle, w, h)
    a_frames = [synth.make("S2", w, h, seed=60 + f) for f in range(2)]
    b_frames = [synth.make("S2", w, h, seed=70 + f) for f in range(2)]
    want_b = [oracle.run(f, s, result_only=True)["result"] for f in b_frames]
    da = [torch.from_numpy(f).to(dev) for f in a_frames]
    db = [torch.from_numpy(f).to(dev) for f in b_frames]
    out = [torch.zeros((h, w), dtype=torch.uint8, device=dev) for _ in range(2)]
    s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    ao = H.component(s, max_batch=2, pipelined=True)
    try:
        ao.set_profiling(True)
        for consumer in (s2, s1):                      # other stream first, then the carrying stream
            ao.prefetch_device([t.data_ptr() for t in db])
            ao.execute_device([t.data_ptr() for t in da], [t.data_ptr() for t in out], s1.cuda_stream)
            s1.synchronize()                           # (makes the cross-stream use legal for the test itself)
            ao.pass_times_ms()                         # fold
            ao.set_profiling(True)                     # restart the window
            ao.execute_device([t.data_ptr() for t in db], [t.data_ptr() for t in out], consumer.cuda_stream)
            consumer.synchronize()
            ms, n = ao.pass_times_ms()
            ran_own_downsample = ms[L.PASS_NAMES.index("downsample")] > 0
            assert ran_own_downsample == (consumer is s2)
            for f in range(2):
                assert np.array_equal(out[f].cpu().numpy(), want_b[f]), (f, consumer is s2)
            ao.set_profiling(True)
    finally:
        ao.close()


def test_failed_resize_leaves_the_context_usable():
    """A meao_resize / first meao_prefetch_batch whose allocation fails returns OUT_OF_MEMORY and the context keeps its size and
    buffers.  The failure is injected through meao_test_fail_next_allocs, which only the `testhooks`
    variant library exports (-DMEAO_TESTING=1; the product ABI has no fault injection): the check runs in its
    own process against that library (tests/resize_failure_check.py)."""
    H.run_against_testhooks("resize_failure_check.py")


def test_the_product_library_has_no_fault_injection(meao_lib):
    """No meao_test_* entry point is exported by the product (they exist in the `testhooks` variant only) and meao_debug_set
    refuses keys it does not know (fault injection was key 5 of ABI <= 4)."""
    from miniengineao_amd import AmbientOcclusion
    assert not hasattr(meao_lib, "meao_test_fail_next_allocs") and not hasattr(meao_lib, "meao_test_pool_refuse_peer")
    with AmbientOcclusion(64, 64) as ao:
        for key in (8, 99, -1):
            with pytest.raises(L.MeaoError) as e:
                ao.debug_set(key, 1)
            assert e.value.status == L.ERR_INVALID_ARGUMENT


@pytest.mark.parametrize("depth_off,out_off", [(4, 0), (0, 1), (8, 2), (0, 0)])
def test_unaligned_device_pointers_take_the_scalar_paths(oracle, depth_off, out_off):
    """Caller pointers that are not 4-texel aligned (include/meao.h: no alignment required)."""
    import torch
    dev = torch.device("cuda", 0)
    w, h = 256, 96                                      # width % 4 == 0: the vector variants would be chosen
    s = H.settings(oracle, w, h)
    depth = synth.make("S2", w, h, seed=17)
    want = oracle.run(depth, s, result_only=True)["result"]
    raw = torch.zeros(w * h * 4 + 64, dtype=torch.uint8, device=dev)
    raw[depth_off:depth_off + w * h * 4] = torch.from_numpy(depth.view(np.uint8).ravel().copy()).to(dev)
    out = torch.zeros(w * h + 64, dtype=torch.uint8, device=dev)
    ao = H.component(s)
    try:
        ao.execute_device([raw.data_ptr() + depth_off], [out.data_ptr() + out_off])
        ao.synchronize()
        got = out[out_off:out_off + w * h].cpu().numpy().reshape(h, w)
        assert np.array_equal(got, want)
        assert not out[out_off + w * h:].any() and not out[:out_off].any()     # nothing written outside
    finally:
        ao.close()


def test_tracing_ranges_can_be_switched_on(oracle):
    w, h = 96, 64
    s = H.settings(oracle, w, h)
    depth = synth.make("S1", w, h)
    ao = H.component(s)
    try:
        try:
            ao.set_tracing(True)
        except L.MeaoError as e:
            assert e.status == L.ERR_UNSUPPORTED        # no libroctx64.so on this box
        assert np.array_equal(ao.render(depth), oracle.run(depth, s, result_only=True)["result"])
        ao.set_tracing(False)
    finally:
        ao.close()


# ---- round 3 ------------------------------------------------------------------------------------------

def test_hostile_frames_mask_in_direct_and_pipelined_calls(oracle):
    import torch
    dev = torch.device("cuda", 0)
    w, h = 200, 120
    s = H.settings(oracle, w, h)
    batches = [[synth.make("S2", w, h, seed=400 + 3 * k + f) for f in range(3)] for k in range(3)]
    batches[1][2] = H.hostile_frame(w, h, 41, density=0.002)
    batches[2][0] = H.hostile_frame(w, h, 42, density=0.002)
    dd = [[torch.from_numpy(f).to(dev) for f in b] for b in batches]
    out = [torch.zeros((h, w), dtype=torch.uint8, device=dev) for _ in range(3)]
    st = torch.cuda.current_stream(dev).cuda_stream
    ao = H.component(s, max_batch=3, pipelined=True)
    try:
        for k, want_mask in enumerate((0, 4, 1)):
            if k + 1 < 3:
                ao.prefetch_device([t.data_ptr() for t in dd[k + 1]])
            ao.execute_device([t.data_ptr() for t in dd[k]], [t.data_ptr() for t in out], st)
            assert ao.hostile_frames() == want_mask, k
    finally:
        ao.close()

