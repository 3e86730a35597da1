"""Runs the REFERENCE'S OWN compute shaders (read from /root/reference, never copied) through
oracle/hlsl_interp.py, driven the way AmbientOcclusion.cs ("AO.cs") drives Unity, and commits
all 17 buffers as fixtures (tests/golden/ref_*.npz).  These are the closest thing to "outputs of
the reference itself" obtainable here: the shader text is executed as written; the host constants
(AO.cs:561-573,660-771) come from the oracle's restatement of the C#, the numerics contract and
resource semantics are those of DESIGN.md section 2.

    python tests/golden/make_reference_goldens.py            # needs /root/reference (build box only)

The fixtures travel to the GPU box; tests/test_reference_goldens.py checks the oracle (CPU) and
the HIP path (GPU) against them bit for bit.
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from miniengineao_amd import synth  # noqa: E402
from oracle import hlsl_interp as HI  # noqa: E402
from oracle import oracle as O  # noqa: E402
from tests import helpers as H  # noqa: E402

SHADERS = "/root/reference/Assets/MiniEngineAO/Shaders"

# name -> (w, h, generator, seed, camera, settings overrides, sky block)
CASES = {
    "ref_s2_37x29_r8": (37, 29, "S2", 21, synth.DEFAULT_CAMERA, {}, False),
    "ref_s2_70x41_f16_rtne_convz": (70, 41, "S2", 22, synth.Camera(reversed_z=False),
                                    dict(ao_format=1, f16_rounding=1, intensity=1.3, thickness_modifier=2.0,
                                         blur_tolerance=-3.0, upsample_tolerance=-6.0, noise_filter_tolerance=-1.5), True),
    "ref_s3_64x48_sponza": (64, 48, "S3", 0, synth.SPONZA_CAMERA, dict(intensity=1.1), False),
}


def textures(s):
    """The RTHandle table of AO.cs:453-475 as interpreter textures over numpy arrays."""
    L = O.lib()
    arrs = O.allocate(s)
    f16 = dict(decode=lambda v: np.float32(L.meao_oracle_f16_to_f32(int(v))),
               encode=lambda v: L.meao_oracle_f32_to_f16(float(v), s.f16_rounding))
    r8 = dict(decode=lambda v: np.float32(L.meao_oracle_unorm8_to_f32(int(v))),
              encode=lambda v: L.meao_oracle_f32_to_unorm8(float(v)))
    ao = r8 if s.ao_format == O.AO_R8 else f16
    tex = {}
    for name, a in arrs.items():
        view = a if a.ndim == 3 else a[None]
        if a.dtype == np.float32:
            tex[name] = HI.Texture(view)
        elif name.startswith(("linear", "tiled")):
            tex[name] = HI.Texture(view, **f16)
        else:
            tex[name] = HI.Texture(view, **ao)
    return arrs, tex


def vec(*vals):
    return ("f", [np.float32(v) for v in vals])


def run_reference_shaders(depth, s, log=print):
    """RebuildCommandBuffers (AO.cs:496-531) over the interpreter."""
    src = {n: open(os.path.join(SHADERS, n + ".compute")).read()
           for n in ("Downsample1", "Downsample2", "Render", "Upsample")}
    arrs, tex = textures(s)
    depth_tex = HI.Texture(np.ascontiguousarray(depth, np.float32)[None])
    dims = [O.level_dims(s.width, s.height, k) for k in range(7)]
    t0 = time.time()

    # ---- PushDownsampleCommands (AO.cs:604-658)
    rev = {"UNITY_REVERSED_Z": "1"} if s.reversed_z else {}
    variants = HI.kernel_variants(src["Downsample1"])
    prog = HI.Parser(HI.lex(HI.preprocess(src["Downsample1"], dict(variants["main"], **rev)))).program()
    prog.funcs["main"].semantics = ["Gid", "GI", "GTid", "DTid"]
    m = HI.Machine(prog)
    m.bind = {"Depth": depth_tex, "LinearZ": tex["linear_depth"], "DS2x": tex["low_depth1"],
              "DS4x": tex["low_depth2"], "DS2xAtlas": tex["tiled_depth1"], "DS4xAtlas": tex["tiled_depth2"]}
    m.const = {"ZBufferParams": vec(*O.zbuffer_params(s))}
    m.dispatch("main", (dims[4][0], dims[4][1], 1))                    # _tiledDepth2 dims (AO.cs:643)
    prog, entry = HI.compile_kernel(src["Downsample2"], "main")
    m = HI.Machine(prog)
    m.bind = {"DS4x": tex["low_depth2"], "DS8x": tex["low_depth3"], "DS16x": tex["low_depth4"],
              "DS8xAtlas": tex["tiled_depth3"], "DS16xAtlas": tex["tiled_depth4"]}
    m.dispatch(entry, (dims[6][0], dims[6][1], 1))                     # _tiledDepth4 dims (AO.cs:657)
    log(f"  downsample done {time.time() - t0:.1f}s")

    # ---- PushRenderCommands (AO.cs:660-748), kernel main_interleaved
    prog, entry = HI.compile_kernel(src["Render"], "main_interleaved")
    for level in range(1, s.num_levels + 1):
        k = O.render_constants(s, level)
        m = HI.Machine(prog)
        m.bind = {"DepthTex": tex[f"tiled_depth{level}"], "Occlusion": tex[f"occlusion{level}"]}
        m.const = {"gInvThicknessTable": [vec(*list(k.inv_thickness)[i:i + 4]) for i in (0, 4, 8)],
                   "gSampleWeightTable": [vec(*list(k.sample_weight)[i:i + 4]) for i in (0, 4, 8)],
                   "gInvSliceDimension": vec(k.inv_slice_dim[0], k.inv_slice_dim[1], 0, 0),
                   "gRejectFadeoff": vec(k.reject_fadeoff), "gIntensity": vec(k.intensity)}
        sw, sh = dims[level + 2]
        m.dispatch(entry, ((sw + 7) // 8, (sh + 7) // 8, 16))          # AO.cs:742-747
        log(f"  render level {level} done {time.time() - t0:.1f}s")

    # ---- PushUpsampleCommands (AO.cs:750-785), generalised to num_levels like the oracle
    lo_ao = tex[f"occlusion{s.num_levels}"]
    for hi in range(s.num_levels - 1, -1, -1):
        kernel = "main" if hi == 0 else "main_blendout"                # AO.cs:758
        prog, entry = HI.compile_kernel(src["Upsample"], kernel)
        k = O.upsample_constants(s, hi + 1)
        m = HI.Machine(prog)
        hi_db = tex["linear_depth"] if hi == 0 else tex[f"low_depth{hi}"]
        dst = tex["result"] if hi == 0 else tex[f"combined{hi}"]
        m.bind = {"LoResDB": tex[f"low_depth{hi + 1}"], "HiResDB": hi_db, "LoResAO1": lo_ao, "AoResult": dst}
        if hi > 0:
            m.bind["HiResAO"] = tex[f"occlusion{hi}"]
        m.const = {"InvLowResolution": vec(k.inv_low_res[0], k.inv_low_res[1], 0, 0),
                   "InvHighResolution": vec(k.inv_high_res[0], k.inv_high_res[1], 0, 0),
                   "NoiseFilterStrength": vec(k.noise_filter_strength), "StepSize": vec(k.step_size),
                   "kBlurTolerance": vec(k.blur_tolerance), "kUpsampleTolerance": vec(k.upsample_tolerance)}
        hw, hh = dims[hi]
        m.dispatch(entry, ((hw + 17) // 16, (hh + 17) // 16, 1))       # AO.cs:782-784
        lo_ao = dst
        log(f"  upsample -> L{hi} done {time.time() - t0:.1f}s")
    return arrs


def make_depth(kind, w, h, seed, cam, sky):
    if kind == "S3":
        depth = synth.atrium(w, h, cam)
    else:
        depth = synth.occluder_field(w, h, seed, n_rects=12, n_discs=12, cam=cam)
    if sky:                                                            # a block of sky texels (1e5)
        depth[h // 3: h // 3 + 9, w // 2:] = 0.0 if cam.reversed_z else 1.0
    return depth


def main(only=None):
    O.build()
    for name, (w, h, kind, seed, cam, over, sky) in CASES.items():
        if only and name not in only:
            continue
        print(name)
        depth = make_depth(kind, w, h, seed, cam, sky)
        s = H.settings(O, w, h, cam=cam, **over)
        ref = run_reference_shaders(depth, s)
        want = O.run(depth, s)
        bad = [k for k in ref if not np.array_equal(ref[k], want[k])]
        print("  interpreter vs oracle: %s" % ("all 17 buffers identical" if not bad else "DIFFER: %s" % bad))
        np.savez_compressed(os.path.join(HERE, name + ".npz"), depth=depth, **ref)


if __name__ == "__main__":
    main(sys.argv[1:])
