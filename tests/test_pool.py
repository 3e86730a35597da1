"""The native multi-device entry point (meao_pool_*): frame f of a batch runs on member f mod G.
gpurun boxes have one GPU, so G > 1 is exercised with several members on device 0 (the dealing,
per-member batching, streams and the peer-copy gather are the same code as with G devices)."""
import numpy as np
import pytest

from miniengineao_amd import _lib as L
from miniengineao_amd import synth
from miniengineao_amd.sharding import frames_for_rank, owner_of_frame
from tests import helpers as H


def test_pool_symbols_are_exported(meao_lib):
    for name in ("meao_pool_create", "meao_pool_destroy", "meao_pool_size", "meao_pool_context",
                 "meao_pool_device_of_frame", "meao_pool_last_error", "meao_pool_set_params",
                 "meao_pool_execute_batch", "meao_pool_gather_to_device", "meao_pool_synchronize",
                 "meao_pool_prefetch_batch", "meao_pool_composite_enqueue", "meao_pool_composite_flush",
                 "meao_pool_gather_path", "meao_pool_configure", "meao_pool_member_placement", "meao_device_numa_node"):
        assert hasattr(meao_lib, name)
    assert meao_lib.meao_pool_size(None) == 0 and meao_lib.meao_pool_device_of_frame(None, 0) == -1


@pytest.mark.gpu
@pytest.mark.parametrize("members", [1, 2, 3])
def test_pool_host_batch_matches_oracle(oracle, members):
    from miniengineao_amd import AmbientOcclusionPool
    w, h, n = 200, 120, 5
    cam = synth.DEFAULT_CAMERA
    s = H.settings(oracle, w, h)
    frames = [synth.make("S2", w, h, seed=500 + f) for f in range(n)]
    with AmbientOcclusionPool(w, h, [0] * members, max_batch=5, near_clip=cam.near, far_clip=cam.far,
                              projection00=cam.proj00(w, h), reversed_z=cam.reversed_z) as pool:
        assert pool.size == members
        assert [pool.device_of_frame(f) for f in range(n)] == [0] * n
        for rep in range(2):
            outs = pool.render_batch(frames)
            for f in range(n):
                assert np.array_equal(outs[f], oracle.run(frames[f], s, result_only=True)["result"]), (rep, f)
        # the member that owns frame f is f mod G: its context holds that frame's intermediates
        lib = L.load()
        for f in range(n):
            m = owner_of_frame(f, members)
            slot = frames_for_rank(n, m, members).index(f)
            ctx = lib.meao_pool_context(pool._pool, m)
            d = L.Desc()
            L.check(lib.meao_get_intermediate(ctx, slot, 10, None, 0, L.MEM_HOST, d))
            occ1 = np.empty((d.height, d.width), np.uint8)
            L.check(lib.meao_get_intermediate(ctx, slot, 10, occ1.ctypes.data, occ1.nbytes, L.MEM_HOST, d), ctx)
            assert np.array_equal(occ1, oracle.run(frames[f], s)["occlusion1"]), f


@pytest.mark.gpu
def test_pool_device_resident_frames_and_gather(oracle):
    torch = pytest.importorskip("torch")
    from miniengineao_amd import AmbientOcclusionPool
    w, h, n = 256, 144, 4
    cam = synth.DEFAULT_CAMERA
    s = H.settings(oracle, w, h)
    frames = [synth.make("S2", w, h, seed=600 + f) for f in range(n)]
    dev = torch.device("cuda", 0)
    dd = [torch.from_numpy(f).to(dev) for f in frames]
    out = [torch.zeros((h, w), dtype=torch.uint8, device=dev) for _ in range(n)]
    gathered = [torch.zeros((h, w), dtype=torch.uint8, device=dev) for _ in range(n)]
    torch.cuda.synchronize(dev)
    with AmbientOcclusionPool(w, h, [0, 0], max_batch=2, near_clip=cam.near, far_clip=cam.far,
                              projection00=cam.proj00(w, h), reversed_z=cam.reversed_z) as pool:
        pool.execute_device([t.data_ptr() for t in dd], [t.data_ptr() for t in out])
        pool.gather_to_device([t.data_ptr() for t in out], [t.data_ptr() for t in gathered], 0)
        pool.synchronize()
        for f in range(n):
            want = oracle.run(frames[f], s, result_only=True)["result"]
            assert np.array_equal(out[f].cpu().numpy(), want), f
            assert np.array_equal(gathered[f].cpu().numpy(), want), f
        with pytest.raises(L.MeaoError):
            pool.execute_device([t.data_ptr() for t in dd] * 2, [t.data_ptr() for t in out] * 2)   # 8 > max_batch * members


@pytest.mark.gpu
def test_gather_takes_the_cross_device_call_when_peer_access_is_refused():
    """VERDICT r5 #6c: on one device meao_pool_gather_to_device only ever takes the same-device copy; the `testhooks` variant
    library can refuse peer access, which sends every copy through hipMemcpyPeerAsync (tests/testhooks_pool_check.py)."""
    H.run_against_testhooks("testhooks_pool_check.py")


@pytest.mark.gpu
def test_pool_worker_placement_and_spin_window(oracle):
    """VERDICT r5 #6b: the workers bind themselves to their device's NUMA node where one is known (a report either way), the spin
    window is a knob, BIND_NUMA is refused once the workers run; results do not depend on any of it."""
    torch = pytest.importorskip("torch")
    import ctypes as C
    from miniengineao_amd import AmbientOcclusionPool
    lib = L.load()
    node, buf = C.c_int32(-7), C.create_string_buffer(4096)
    assert lib.meao_device_numa_node(0, C.byref(node), buf, len(buf)) == L.OK and node.value >= -1
    assert lib.meao_device_numa_node(99, C.byref(node), None, 0) == L.ERR_INVALID_ARGUMENT
    w, h, n = 200, 120, 4
    cam = synth.DEFAULT_CAMERA
    s = H.settings(oracle, w, h)
    frames = [synth.make("S2", w, h, seed=520 + f) for f in range(n)]
    dev = torch.device("cuda", 0)
    dd = [torch.from_numpy(f).to(dev) for f in frames]
    out = [torch.zeros((h, w), dtype=torch.uint8, device=dev) for _ in range(n)]
    torch.cuda.synchronize(dev)
    for bind, spin in ((1, 0), (0, 100), (1, 2000)):
        with AmbientOcclusionPool(w, h, [0, 0], max_batch=2, near_clip=cam.near, far_clip=cam.far,
                                  projection00=cam.proj00(w, h), reversed_z=cam.reversed_z) as pool:
            pool.configure(L.POOL_BIND_NUMA, bind)
            pool.configure(L.POOL_SPIN_US, spin)
            with pytest.raises(L.MeaoError):
                pool.configure(L.POOL_SPIN_US, -1)
            for _ in range(3):
                pool.execute_device([t.data_ptr() for t in dd], [t.data_ptr() for t in out])
            pool.synchronize()
            for m in range(2):
                place = pool.member_placement(m)
                assert place["numa_node"] == node.value or node.value < 0
                assert place["worker_bound"] in (False, True) and (bind or not place["worker_bound"])
                if place["numa_node"] < 0:
                    assert not place["worker_bound"]            # nothing to bind to where the kernel knows no node
            with pytest.raises(L.MeaoError) as e:               # the workers are running now
                pool.configure(L.POOL_BIND_NUMA, 0)
            assert e.value.status == L.ERR_UNSUPPORTED
            pool.configure(L.POOL_SPIN_US, 50)                  # the spin window can change at any time
            pool.execute_device([t.data_ptr() for t in dd], [t.data_ptr() for t in out])
            pool.synchronize()
            for f in range(n):
                assert np.array_equal(out[f].cpu().numpy(), oracle.run(frames[f], s, result_only=True)["result"]), (bind, spin, f)


@pytest.mark.gpu
@pytest.mark.parametrize("members", [2, 3])
def test_pool_pipelined_step_is_bit_exact(oracle, members):
    """meao_pool_prefetch_batch: every member's execute carries its share of the NEXT batch's downsample pass;
    three consecutive batches (distinct frames, one hostile), every output against the oracle; the members'
    contexts must actually have skipped their own downsample pass in steps 2 and 3."""
    torch = pytest.importorskip("torch")
    from miniengineao_amd import AmbientOcclusionPool
    w, h, n = 320, 192, 5
    cam = synth.DEFAULT_CAMERA
    s = H.settings(oracle, w, h)
    batches = [[synth.make("S2", w, h, seed=700 + 10 * k + f) for f in range(n)] for k in range(3)]
    batches[1][3] = H.hostile_frame(w, h, 71, density=0.002)
    dev = torch.device("cuda", 0)
    dd = [[torch.from_numpy(f).to(dev) for f in b] for b in batches]
    out = [[torch.zeros((h, w), dtype=torch.uint8, device=dev) for _ in range(n)] for _ in range(3)]
    torch.cuda.synchronize(dev)
    lib = L.load()
    with AmbientOcclusionPool(w, h, [0] * members, max_batch=3, near_clip=cam.near, far_clip=cam.far,
                              projection00=cam.proj00(w, h), reversed_z=cam.reversed_z, pipelined=True) as pool:
        for m in range(members):
            L.check(lib.meao_set_profiling(pool.member_context(m), 1))
        for k in range(3):
            if k + 1 < 3:
                pool.prefetch_device([t.data_ptr() for t in dd[k + 1]])
            pool.execute_device([t.data_ptr() for t in dd[k]], [t.data_ptr() for t in out[k]])
        pool.synchronize()
        for k in range(3):
            for f in range(n):
                want = oracle.run(batches[k][f], s, result_only=True)["result"]
                ok, bad = H.nan_aware_equal(out[k][f].cpu().numpy(), want)
                assert ok, (k, f, int(bad.sum()))
        import ctypes as C
        for m in range(members):
            ms, cnt = (C.c_float * L.NUM_PASSES)(), C.c_int32()
            L.check(lib.meao_get_pass_times(pool.member_context(m), C.byref(ms), C.byref(cnt)))
            assert cnt.value == 3 and ms[0] > 0          # three executes; the downsample pass ran (only in the first)
        assert pool.gather_path(0, 0) == L.POOL_PATH_SAME_DEVICE
        assert pool.gather_path(members, 0) < 0 and pool.gather_path(0, 99) < 0


@pytest.mark.gpu
def test_pool_composite_rides_in_the_owning_members_render_kernel(oracle):
    torch = pytest.importorskip("torch")
    from miniengineao_amd import AmbientOcclusionPool
    w, h, n = 192, 128, 4
    cam = synth.DEFAULT_CAMERA
    s = H.settings(oracle, w, h)
    frames = [synth.make("S2", w, h, seed=800 + f) for f in range(n)]
    want_ao = [oracle.run(f, s, result_only=True)["result"] for f in frames]
    base = (np.random.default_rng(9).random((h, w, 4)) * 2.0).astype(np.float16).view(np.uint16)
    dev = torch.device("cuda", 0)
    dd = [torch.from_numpy(f).to(dev) for f in frames]
    out = [torch.zeros((h, w), dtype=torch.uint8, device=dev) for _ in range(n)]
    cols = [torch.from_numpy(base.view(np.int16).copy()).to(dev) for _ in range(2 * n)]
    torch.cuda.synchronize(dev)
    with AmbientOcclusionPool(w, h, [0, 0], max_batch=2, near_clip=cam.near, far_clip=cam.far,
                              projection00=cam.proj00(w, h), reversed_z=cam.reversed_z) as pool:
        dp, op = [t.data_ptr() for t in dd], [t.data_ptr() for t in out]
        pool.execute_device(dp, op)
        pool.composite_enqueue_device(L.COMPOSITE_MULTIPLY, op, [t.data_ptr() for t in cols[:n]])
        pool.execute_device(dp, op)                                     # carries the composite of the first call
        pool.composite_enqueue_device(L.COMPOSITE_MULTIPLY, op, [t.data_ptr() for t in cols[n:]])
        pool.composite_flush()                                          # plain launches for what still waits
        pool.synchronize()
        for f in range(n):
            want = base.copy()
            oracle.composite(want_ao[f], want, 0)
            assert np.array_equal(out[f].cpu().numpy(), want_ao[f]), f
            assert np.array_equal(cols[f].cpu().numpy().view(np.uint16), want), (f, "carried")
            assert np.array_equal(cols[n + f].cpu().numpy().view(np.uint16), want), (f, "flushed")


@pytest.mark.gpu
def test_pool_calls_leave_the_current_device_alone_and_destroy_with_work_in_flight(oracle):
    """members with launches in flight and a waiting composite are destroyed in the order
    context first, stream second; pool entry points restore the calling thread's device."""
    torch = pytest.importorskip("torch")
    import ctypes as C
    lib = L.load()
    w, h, n = 160, 96, 4
    cam = synth.DEFAULT_CAMERA
    dev = torch.device("cuda", 0)
    frames = [synth.make("S2", w, h, seed=900 + f) for f in range(n)]
    dd = [torch.from_numpy(f).to(dev) for f in frames]
    out = [torch.zeros((h, w), dtype=torch.uint8, device=dev) for _ in range(n)]
    col = [torch.zeros((h, w, 4), dtype=torch.float16, device=dev) for _ in range(n)]
    torch.cuda.synchronize(dev)
    cfg = L.Config()
    lib.meao_default_config(C.byref(cfg))
    cfg.width, cfg.height, cfg.max_batch = w, h, 2
    pool = C.c_void_p()
    L.check(lib.meao_pool_create(C.byref(cfg), (C.c_int32 * 2)(0, 0), 2, C.byref(pool)))
    prm = L.Params()
    lib.meao_default_params(C.byref(prm))
    prm.near_clip, prm.far_clip, prm.proj00, prm.reversed_z = cam.near, cam.far, cam.proj00(w, h), 1 if cam.reversed_z else 0
    L.check(lib.meao_pool_set_params(pool, C.byref(prm)))
    dp, op = (C.c_void_p * n)(*[t.data_ptr() for t in dd]), (C.c_void_p * n)(*[t.data_ptr() for t in out])
    for _ in range(3):       # launches in flight
        L.check(lib.meao_pool_execute_batch(pool, n, dp, L.MEM_DEVICE, op, L.MEM_DEVICE))
    L.check(lib.meao_pool_composite_enqueue(pool, L.COMPOSITE_MULTIPLY, n, op, (C.c_void_p * n)(*[t.data_ptr() for t in col]), None))
    assert torch.cuda.current_device() == 0
    assert lib.meao_pool_destroy(pool) == L.OK
    torch.cuda.synchronize(dev)
    s = H.settings(oracle, w, h)
    for f in range(n):
        assert np.array_equal(out[f].cpu().numpy(), oracle.run(frames[f], s, result_only=True)["result"]), f
        assert float(col[f].abs().max()) == 0.0          # the waiting composite was discarded, not run on a dead stream
