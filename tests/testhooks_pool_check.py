"""Body of tests/test_pool.py::test_gather_takes_the_cross_device_call_when_peer_access_is_refused.  Runs in its own process against
the `testhooks` variant library (MEAO_LIB_PATH; -DMEAO_TESTING=1), which exports meao_test_pool_refuse_peer: every copy of
meao_pool_gather_to_device then goes through hipMemcpyPeerAsync and meao_pool_gather_path reports STAGED, as on a node whose
devices offer no peer access -- the branch a one-GPU box otherwise never takes (VERDICT r5 #6c).  Exit code 0 = passed."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from miniengineao_amd import AmbientOcclusionPool
from miniengineao_amd import _lib as L
from miniengineao_amd import synth
from oracle import oracle
from tests import helpers as H


def main():
    lib = L.load()
    refuse = lib.meao_test_pool_refuse_peer          # AttributeError unless this is the testhooks build
    refuse.restype, refuse.argtypes = C.c_int32, [C.c_void_p, C.c_int32]
    oracle.build()
    w, h, n = 256, 144, 4
    cam = synth.DEFAULT_CAMERA
    s = H.settings(oracle, w, h)
    frames = [synth.make("S2", w, h, seed=650 + f) for f in range(n)]
    dev = torch.device("cuda", 0)
    dd = [torch.from_numpy(f).to(dev) for f in frames]
    out = [torch.zeros((h, w), dtype=torch.uint8, device=dev) for _ in range(n)]
    gathered = [torch.zeros((h, w), dtype=torch.uint8, device=dev) for _ in range(n)]
    torch.cuda.synchronize(dev)
    with AmbientOcclusionPool(w, h, [0, 0], max_batch=2, near_clip=cam.near, far_clip=cam.far,
                              projection00=cam.proj00(w, h), reversed_z=cam.reversed_z) as pool:
        assert pool.gather_path(0, 0) == L.POOL_PATH_SAME_DEVICE and pool.gather_path(1, 0) == L.POOL_PATH_SAME_DEVICE
        assert refuse(pool._pool, 1) == L.OK
        assert pool.gather_path(0, 0) == L.POOL_PATH_STAGED and pool.gather_path(1, 0) == L.POOL_PATH_STAGED
        for rep in range(2):
            pool.execute_device([t.data_ptr() for t in dd], [t.data_ptr() for t in out])
            pool.gather_to_device([t.data_ptr() for t in out], [t.data_ptr() for t in gathered], 0)      # ordered behind the producers
            pool.synchronize()
            for f in range(n):
                want = oracle.run(frames[f], s, result_only=True)["result"]
                assert np.array_equal(out[f].cpu().numpy(), want), (rep, f)
                assert np.array_equal(gathered[f].cpu().numpy(), want), (rep, f, "gathered through the cross-device call")
            for t in gathered:
                t.zero_()
        assert refuse(pool._pool, 0) == L.OK and pool.gather_path(0, 0) == L.POOL_PATH_SAME_DEVICE
    print("testhooks pool check ok")


if __name__ == "__main__":
    main()
