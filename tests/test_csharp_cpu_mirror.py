"""bindings/csharp/AmbientOcclusionCpu.cs (the scalar C# CPU implementation north_star asks for)
cannot be compiled in this image, so it is EXECUTED by oracle/csharp_interp.py and compared with
the C oracle buffer by buffer, bit for bit."""
import math
import os
import struct

import numpy as np
import pytest

from miniengineao_amd import synth
from oracle import csharp_interp as CS
from oracle import hlsl_interp as HI
from oracle import oracle as O
from tests import helpers as H

F = np.float32
SRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bindings", "csharp", "AmbientOcclusionCpu.cs")


class _NS:
    def __init__(self, **kw):
        self.__dict__.update(kw)


def _globals():
    bits = lambda x: struct.unpack("<i", struct.pack("<f", float(x)))[0]
    unbits = lambda i: F(struct.unpack("<f", struct.pack("<I", int(i) & 0xFFFFFFFF))[0])
    return {
        "MathF": _NS(FusedMultiplyAdd=HI.fmaf, Abs=lambda v: F(abs(v))),
        "Math": _NS(Sqrt=lambda v: math.sqrt(float(v)), Pow=lambda a, b: math.pow(float(a), float(b))),
        "BitConverter": _NS(SingleToInt32Bits=bits, Int32BitsToSingle=unbits),
    }


def run_mirror(depth, s):
    classes = CS.load(SRC)
    it = CS.Interp(classes, _globals())
    cls = classes["AmbientOcclusionCpu"]
    ao = it.new_instance(cls, [])
    for field, value in (("noiseFilterTolerance", s.noise_filter_tolerance), ("blurTolerance", s.blur_tolerance),
                         ("upsampleTolerance", s.upsample_tolerance), ("thicknessModifier", s.thickness_modifier),
                         ("intensity", s.intensity), ("nearClipPlane", s.near_clip), ("farClipPlane", s.far_clip),
                         ("projection00", s.proj00)):
        assert field in ao.f
        ao.f[field] = F(value)
    ao.f["usesReversedZBuffer"] = bool(s.reversed_z)
    flat = [F(v) for v in np.asarray(depth, np.float32).ravel()]
    with np.errstate(all="ignore"):
        it.call_method(ao, cls, "Run", [flat, s.width, s.height])
    dims = lambda k: ((s.width + (1 << k) - 1) >> k, (s.height + (1 << k) - 1) >> k)
    out = {"linear_depth": np.array(ao.f["linearDepth"], np.uint16).reshape(s.height, s.width),
           "result": np.array(ao.f["result"], np.uint8).reshape(s.height, s.width)}
    for k in range(1, 5):
        w, h = dims(k)
        tw, th = dims(k + 2)
        out[f"low_depth{k}"] = np.array(ao.f["lowDepth"][k - 1], np.float32).reshape(h, w)
        out[f"tiled_depth{k}"] = np.array(ao.f["tiledDepth"][k - 1], np.uint16).reshape(16, th, tw)
        out[f"occlusion{k}"] = np.array(ao.f["occlusion"][k - 1], np.uint8).reshape(h, w)
        if k <= 3:
            out[f"combined{k}"] = np.array(ao.f["combined"][k - 1], np.uint8).reshape(h, w)
    return out


CASES = {
    "reversed_z": (26, 19, synth.DEFAULT_CAMERA, {}),
    "conventional_z_tuned": (21, 23, synth.Camera(reversed_z=False),
                             dict(intensity=1.4, thickness_modifier=2.5, blur_tolerance=-3.0,
                                  upsample_tolerance=-7.0, noise_filter_tolerance=-1.0)),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_csharp_mirror_matches_oracle(name):
    w, h, cam, over = CASES[name]
    depth = synth.occluder_field(w, h, 5, n_rects=6, n_discs=6, cam=cam)
    depth[2:5, 3:9] = 0.0 if cam.reversed_z else 1.0          # sky texels
    s = H.settings(O, w, h, cam=cam, **over)
    want = O.run(depth, s)
    got = run_mirror(depth, s)
    assert sorted(got) == sorted(k for k in want if k != "depth")
    bad = [k for k in got if not np.array_equal(got[k].view(np.uint8) if got[k].dtype == np.float32 else got[k],
                                                want[k].view(np.uint8) if want[k].dtype == np.float32 else want[k])]
    assert not bad, bad


def test_interpreter_extensions_used_by_the_mirror():
    src = """
    class T {
        static int Hex() { return 0x7bff | (0x10 << 4); }
        static int Casts(float x) { return (byte)(int)x + (ushort)70000; }
        static int Jagged() { float[][] a = new float[3][]; a[1] = new float[2]; return a.Length * 10 + a[1].Length; }
        static float Init() { float[] c = { 1, 2, 3.5f }; return c[0] / c[1]; }
    }"""
    classes = CS.Parser(CS.lex(CS.strip_noise(src))).compilation_unit()
    it = CS.Interp(classes, {})
    call = lambda n, *a: it.call_method(None, classes["T"], n, list(a))
    assert call("Hex") == 0x7bff | 0x100
    assert call("Casts", F(300.7)) == (300 & 0xFF) + (70000 & 0xFFFF)
    assert call("Jagged") == 32
    r = call("Init")
    assert isinstance(r, np.float32) and r == F(0.5)
