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
                 "meao_pool_execute_batch", "meao_pool_gather_to_device", "meao_pool_synchronize"):
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
