"""Fixtures for the composite and the de-tile debug view, produced by executing the raster passes of
the reference's Blit.shader from their source text (oracle/shaderlab_interp.py; the text is read from
/root/reference at generation time and never copied here).

    python tests/golden/make_blit_goldens.py      ->  tests/golden/ref_blit_passes.npz

Pass -> host call site in the reference: 1 = deferred ambient-only composite (AO.cs:830-834, MRT
GBuffer0 + camera target), 2 = standard composite (AO.cs:837), 3 = debug view of the AO texture
(AO.cs:826), 4 = de-tile view of a 16-slice array (AO.cs:811-813).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import oracle as O                 # noqa: E402  (storage conversions only)
from oracle import shaderlab_interp as SL      # noqa: E402

BLIT = "/root/reference/Assets/MiniEngineAO/Shaders/Blit.shader"
OUT = os.path.join(HERE, "ref_blit_passes.npz")


def f16_bits(x):
    L = O.lib()
    return np.vectorize(lambda v: L.meao_oracle_f32_to_f16(float(v), O.F16_RTNE), otypes=[np.uint16])(x)


def unorm8(x):
    L = O.lib()
    return np.vectorize(lambda v: L.meao_oracle_f32_to_unorm8(float(v)), otypes=[np.uint8])(x)


def inputs(w, h, seed):
    rng = np.random.default_rng(seed)
    ao = rng.integers(0, 256, (h, w), dtype=np.uint8)
    ao[0, :4] = [0, 255, 1, 254]
    color = (rng.random((h, w, 4)) * 4.0 - 0.25).astype(np.float32)        # HDR: > 1 and a few negatives
    color[1, :3, :] = [[0.0] * 4, [65504.0] * 4, [1e-7] * 4]
    color16 = f16_bits(color)
    gbuf = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
    return ao, color16, gbuf


def run_composite_pass(p, ao, color16, gbuf):
    """Returns (color16', gbuffer0') after pass `p` with the blend of its ShaderLab text."""
    h, w = ao.shape
    ao_f = (ao.astype(np.float32) / np.float32(255)).astype(np.float32)      # UNORM8 sampled as float
    frag = SL.fragment_outputs(p, {"_AOTexture": ao_f}, w, h)
    color = O.f16_bits_to_f32(color16)
    gb = (gbuf.astype(np.float32) / np.float32(255)).astype(np.float32)
    out_c, out_g = np.zeros_like(color), np.zeros_like(gb)
    # MRT of pass 1: SV_Target0 = GBuffer0, SV_Target1 = the camera target (AO.cs:832); single target otherwise
    src_color = frag.get("gbuffer3", frag.get("SV_Target"))
    src_gbuf = frag.get("gbuffer0")
    for y in range(h):
        for x in range(w):
            out_c[y, x] = SL.blend(p["blend"], src_color[y, x], color[y, x])
            if src_gbuf is not None:
                out_g[y, x] = SL.blend(p["blend"], src_gbuf[y, x], gb[y, x])
    return f16_bits(out_c), (unorm8(out_g) if src_gbuf is not None else gbuf)


def main():
    O.build()
    text = open(BLIT).read()
    ps = SL.passes(text)
    assert len(ps) == 5 and ps[3]["name"] == "Debug" and ps[4]["name"] == "Detile", [p["name"] for p in ps]
    w, h = 28, 18
    ao, color16, gbuf = inputs(w, h, 7)
    out = {"ao": ao, "color_in": color16, "gbuffer0_in": gbuf,
           "blend_pass1": np.array(ps[1]["blend"]), "blend_pass2": np.array(ps[2]["blend"])}
    for k, name in ((2, "multiply"), (1, "ambient_only"), (3, "debug")):
        c, g = run_composite_pass(ps[k], ao, color16, gbuf)
        out[f"color_{name}"], out[f"gbuffer0_{name}"] = c, g
        print(f"pass {k} ({name}): blend = {ps[k]['blend']}")
    # pass 4: the 4 x 4 slice grid of a tiled array, sampled at the pixel centres of a 52 x 36 target
    rng = np.random.default_rng(11)
    tiled16 = f16_bits((rng.random((16, 5, 7)) * 2.0).astype(np.float32))
    tw, th = 52, 36
    frag = SL.fragment_outputs(ps[4], {"_TileTexture": O.f16_bits_to_f32(tiled16)}, tw, th)
    out["tiled_in"] = tiled16
    out["detile_r"] = frag["SV_Target"][:, :, 0].copy()
    np.savez_compressed(OUT, **out)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
