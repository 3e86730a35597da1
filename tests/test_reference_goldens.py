"""Fixtures produced by executing the REFERENCE'S OWN shader source (tests/golden/
make_reference_goldens.py, oracle/hlsl_interp.py).  CPU: both oracle restatements reproduce them;
where /root/reference is present the shaders are re-interpreted live on a tiny frame.  GPU: the HIP
path reproduces them without any oracle in the loop."""
import os

import numpy as np
import pytest

from tests import helpers as H
from tests.golden import make_reference_goldens as R

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _case(oracle, name):
    w, h, kind, seed, cam, over, sky = R.CASES[name]
    fx = np.load(os.path.join(HERE, name + ".npz"))
    return fx, H.settings(oracle, w, h, cam=cam, **over)


@pytest.mark.parametrize("name", sorted(R.CASES))
def test_oracle_matches_reference_shader_outputs(oracle, name):
    fx, s = _case(oracle, name)
    w, h, kind, seed, cam, over, sky = R.CASES[name]
    assert np.array_equal(R.make_depth(kind, w, h, seed, cam, sky), fx["depth"])
    for emulate in (False, True):
        out = oracle.run(fx["depth"], s, emulate_hlsl=emulate)
        for key, arr in out.items():
            assert np.array_equal(arr, fx[key]), H.diff_report(key, arr, fx[key])


@pytest.mark.skipif(not os.path.isdir(R.SHADERS), reason="reference checkout not present (GPU box)")
def test_reference_shaders_interpreted_live(oracle):
    """Re-run the reference's .compute files through the interpreter now (not from fixtures)."""
    from miniengineao_amd import synth
    w, h = 18, 11
    cam = synth.Camera(reversed_z=False)
    depth = R.make_depth("S2", w, h, 5, cam, True)
    s = H.settings(oracle, w, h, cam=cam, num_levels=2, intensity=0.8, thickness_modifier=3.0)
    ref = R.run_reference_shaders(depth, s, log=lambda *_: None)
    want = oracle.run(depth, s)
    for i in H.valid_debug_ids(2):
        assert np.array_equal(ref[H.NAMES[i]], want[H.NAMES[i]]), H.NAMES[i]


def test_interpreter_semantics():
    """The pieces of HLSL the shaders lean on: C octal literals, 32-bit unsigned wrap-around,
    int/uint/float promotion, swizzles, mad contraction, D3D NaN rules."""
    from oracle import hlsl_interp as HI
    src = """
    RWTexture2D<float> Out;
    float helper(uint a, int b) { return a + b; }
    [numthreads(1, 1, 1)]
    void main(uint3 DTid : SV_DispatchThreadID)
    {
        uint2 p = DTid.xy + DTid.xy - 2;            // wraps to 0xFFFFFFFE
        int2 q = int2(p);                           // reinterpreted as -2
        float4 v = float4(1, 2, 3, 4);
        Out[uint2(0, 0)] = (011 == 9) ? 1.0 : 0.0;
        Out[uint2(1, 0)] = q.x;
        Out[uint2(2, 0)] = v.wzyx.y + dot(v, 1);    // 3 + 10
        Out[uint2(3, 0)] = 1.000244140625 * 1.000244140625 - 1.00048828125;   // fused: 2^-24, unfused: 0
        Out[uint2(4, 0)] = saturate(0.0 / 0.0);     // NaN -> 0
        Out[uint2(5, 0)] = helper(7, -9);           // uint + int -> uint, then to float
        Out[q + int2(1, 2)] = 5.0;                  // (-1, 0): dropped
        Out[uint2(6, 0)] = (5 & 011) | (1 << 4);    // 1 | 16
        Out[uint2(7, 0)] = lerp(1, 0.25, 2.0);      // 1 + 2*(0.25-1)
    }
    """
    prog = HI.Parser(HI.lex(HI.preprocess(src, {}))).program()
    prog.funcs["main"].semantics = ["DTid"]
    m = HI.Machine(prog)
    out = np.full((1, 1, 8), -1.0, np.float32)
    m.bind = {"Out": HI.Texture(out)}
    m.dispatch("main", (1, 1, 1))
    assert out[0, 0].tolist() == [1.0, -2.0, 13.0, 2.0 ** -24, 0.0, 4294967296.0, 17.0, -0.5]


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(R.CASES))
def test_gpu_matches_reference_shader_outputs(name):
    from oracle import oracle as O      # Settings container only
    fx, s = _case(O, name)
    ao = H.component(s)
    try:
        got = ao.render(fx["depth"])
        assert np.array_equal(got, fx["result"]), H.diff_report("result", got, fx["result"])
        for i in H.valid_debug_ids(s.num_levels):
            assert np.array_equal(ao.debug_buffer(i), fx[H.NAMES[i]]), H.NAMES[i]
    finally:
        ao.close()
