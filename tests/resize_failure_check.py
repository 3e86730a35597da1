"""Body of tests/test_gpu_more.py::test_failed_resize_leaves_the_context_usable.  Runs in its own process against the
`testhooks` variant library (MEAO_LIB_PATH; built with -DMEAO_TESTING=1), the only build that exports the fault injection
meao_test_fail_next_allocs -- the product library has no such entry point.  Exit code 0 = passed."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from miniengineao_amd import _lib as L
from miniengineao_amd import synth
from oracle import oracle
from tests import helpers as H


def raises(status, fn, *args):
    try:
        fn(*args)
    except L.MeaoError as e:
        assert e.status == status, (e.status, status)
        return
    raise AssertionError("no error raised")


def main():
    lib = L.load()
    inject = lib.meao_test_fail_next_allocs          # AttributeError unless this is the testhooks build
    inject.restype, inject.argtypes = C.c_int32, [C.c_void_p, C.c_int32]
    oracle.build()
    w, h = 160, 90
    s = H.settings(oracle, w, h)
    depth = synth.make("S2", w, h, seed=5)
    want = oracle.run(depth, s, result_only=True)["result"]
    ao = H.component(s, max_batch=2)
    try:
        assert np.array_equal(ao.render(depth), want)
        assert inject(ao._ctx, 2) == L.OK
        raises(L.ERR_OUT_OF_MEMORY, ao.resize, 640, 360)
        assert (ao.width, ao.height) == (w, h)
        assert np.array_equal(ao.render(depth), want)
        d = torch.from_numpy(depth).cuda()
        raises(L.ERR_OUT_OF_MEMORY, ao.prefetch_device, [d.data_ptr()])      # first announcement needs the second downsample set
        assert np.array_equal(ao.render(depth), want)
        assert inject(ao._ctx, 0) == L.OK
        raises(L.ERR_INVALID_ARGUMENT, ao.resize, 0, 10)
        ao.resize(96, 64)                               # a resize that fits still works afterwards
        d2 = synth.make("S1", 96, 64)
        s2 = H.settings(oracle, 96, 64)
        s2.proj00 = s.proj00                            # the camera did not change
        assert np.array_equal(ao.render(d2), oracle.run(d2, s2, result_only=True)["result"])
    finally:
        ao.close()
    print("resize failure check ok")


if __name__ == "__main__":
    main()
