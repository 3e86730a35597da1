"""Committed golden fixtures (generated from the oracle by tests/golden/make_golden.py; the
reference itself has no goldens).  CPU: the oracle still reproduces them.  GPU: the HIP path
reproduces them without the oracle in the loop."""
import json
import os

import numpy as np
import pytest

from tests import helpers as H
from tests.golden import make_golden as G

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
with open(os.path.join(HERE, "golden_checksums.json")) as _f:
    SUMS = json.load(_f)


@pytest.mark.parametrize("name", sorted(G.SMALL))
def test_oracle_reproduces_small_fixtures(oracle, name):
    w, h, kind, seed, cam, over = G.SMALL[name]
    fx = np.load(os.path.join(HERE, name + ".npz"))
    depth = G.make_depth(kind, w, h, seed, cam)
    assert np.array_equal(depth, fx["depth"]), "synthetic input generator drifted"
    out = oracle.run(fx["depth"], H.settings(oracle, w, h, cam=cam, **over))
    for key, arr in out.items():
        assert np.array_equal(arr, fx[key]), H.diff_report(key, arr, fx[key])
        assert H.checksum(arr) == SUMS[name][key]


@pytest.mark.parametrize("name", sorted(G.LARGE))
def test_oracle_reproduces_full_size_checksums(oracle, name):
    w, h, kind, seed, cam, over = G.LARGE[name]
    depth = G.make_depth(kind, w, h, seed, cam)
    assert H.checksum(depth) == SUMS[name]["depth"]
    out = oracle.run(depth, H.settings(oracle, w, h, cam=cam, **over), nthreads=oracle.host_cores(),
                     result_only=True)["result"]
    assert H.checksum(out) == SUMS[name]["result"]


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(G.SMALL))
def test_gpu_reproduces_small_fixtures(name):
    w, h, kind, seed, cam, over = G.SMALL[name]
    fx = np.load(os.path.join(HERE, name + ".npz"))
    from oracle import oracle as O      # only for the Settings container
    s = H.settings(O, w, h, cam=cam, **over)
    ao = H.component(s)
    try:
        got = ao.render(fx["depth"])
        assert np.array_equal(got, fx["result"]), H.diff_report("result", got, fx["result"])
        for i in H.valid_debug_ids(s.num_levels, s.hq_levels):
            assert np.array_equal(ao.debug_buffer(i), fx[H.NAMES[i]]), H.NAMES[i]
    finally:
        ao.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(G.LARGE))
def test_gpu_reproduces_full_size_checksums(name):
    w, h, kind, seed, cam, over = G.LARGE[name]
    from oracle import oracle as O
    s = H.settings(O, w, h, cam=cam, **over)
    depth = G.make_depth(kind, w, h, seed, cam)
    assert H.checksum(depth) == SUMS[name]["depth"]
    ao = H.component(s)
    try:
        assert H.checksum(ao.render(depth)) == SUMS[name]["result"]
    finally:
        ao.close()
