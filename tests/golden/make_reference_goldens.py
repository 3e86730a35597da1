"""Runs THE REFERENCE ITSELF from its source text (read from /root/reference, never copied):
AmbientOcclusion.cs is executed by oracle/csharp_interp.py against recording Unity mocks
(tests/golden/unity_mocks.py) and yields the render-texture allocations, the ten compute
dispatches, their texture bindings, constant blocks and group counts; each dispatch then runs the
reference's .compute source through oracle/hlsl_interp.py.  All 17 buffers are committed as
fixtures (tests/golden/ref_*.npz).  What is NOT taken from the reference, because its platform
would supply it: the numerics contract and resource/format semantics of DESIGN.md section 2, and
Unity's API behaviour (mocked).

    python tests/golden/make_reference_goldens.py            # needs /root/reference (build box only)

The fixtures travel to the GPU box; tests/test_reference_goldens.py checks the oracle (CPU) and
the HIP path (GPU) against them bit for bit.

Variants the reference carries but its host never dispatches are pinned at the
shader level: Render.main (wide), SAMPLE_EXHAUSTIVELY and Upsample.main_premin* run from the
reference's source text; their constants come from the reference's own PushRenderCommands /
PushUpsampleCommands (called through the C# interpreter with a non-tiled source where needed).
What the reference does NOT contain and is therefore this project's wiring (following the
Microsoft MiniEngine original): which levels get a Render.main pass and that its output is the
LoResAO2 of the next upsample; the un-zeroed weight table of the exhaustive set (AO.cs:709 FIXME).
Single-pass stereo IS a reference host path (AO.cs:392-401,680) and is recorded from the C#.
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
    "ref_stereo_2x24x27": (48, 27, "S2", 23, synth.DEFAULT_CAMERA, dict(single_pass_stereo=True), False),
    "ref_hq4_45x31": (45, 31, "S2", 24, synth.DEFAULT_CAMERA, dict(hq_levels=4, intensity=1.2), True),
    "ref_hq2_exhaustive_52x38_f16": (52, 38, "S2", 25, synth.Camera(reversed_z=False),
                                     dict(hq_levels=2, sample_set=1, ao_format=1, thickness_modifier=1.5), False),
    "ref_s2_37x29_r8": (37, 29, "S2", 21, synth.DEFAULT_CAMERA, {}, False),
    "ref_s2_70x41_f16_rtne_convz": (70, 41, "S2", 22, synth.Camera(reversed_z=False),
                                    dict(ao_format=1, f16_rounding=1, intensity=1.3, thickness_modifier=2.0,
                                         blur_tolerance=-3.0, upsample_tolerance=-6.0, noise_filter_tolerance=-1.5), True),
    "ref_s3_64x48_sponza": (64, 48, "S3", 0, synth.SPONZA_CAMERA, dict(intensity=1.1), False),
    # Round 5: frames whose every level spans several HIP tiles (render 128x32 / 64x16, upsample 64x64 /
    # 64x32, downsample 128x32), with interior (whole-tile) and edge tiles; minutes of interpreter time each.
    "ref_s2_322x182_r8": (322, 182, "S2", 31, synth.DEFAULT_CAMERA, {}, False),
    "ref_s2_644x364_f16_rtne_convz_sky": (644, 364, "S2", 32, synth.Camera(reversed_z=False),
                                          dict(ao_format=1, f16_rounding=1, intensity=1.15, thickness_modifier=1.7), True),
    "ref_s2h_516x260_hostile_r8": (516, 260, "S2H", 33, synth.DEFAULT_CAMERA, {}, False),
    # storage conversions by the independent NumPy model below instead of the oracle's C functions
    "ref_s2_150x86_r8_numpy_codecs": (150, 86, "S2", 34, synth.DEFAULT_CAMERA, dict(intensity=0.9), True),
    "ref_s2_134x70_f16_rtz_numpy_codecs": (134, 70, "S2", 35, synth.Camera(reversed_z=False),
                                           dict(ao_format=1, thickness_modifier=2.5), False),
    "ref_s3_1920x1080_sponza_r8_checksums": (1920, 1080, "S3", 0, synth.SPONZA_CAMERA, dict(intensity=1.1), False),
    # BASELINE config 3's size, the headline workload (4K S2): 4-5 hours of interpreter time
    "ref_s2_3840x2160_r8_checksums": (3840, 2160, "S2", 41, synth.DEFAULT_CAMERA, {}, True),
    # Round 6: fp16 AO storage -- BASELINE config 5's storage mode, AO.cs:466-475 format table with RHalf AO targets --
    # at a BASELINE-class size, in both f16 store roundings (RTZ = the canonical one the bench runs; RTNE with a conventional Z buffer)
    "ref_s2_1920x1080_f16_rtz_checksums": (1920, 1080, "S2", 51, synth.DEFAULT_CAMERA, dict(ao_format=1), True),
    "ref_s2_1920x1080_f16_rtne_convz_checksums": (1920, 1080, "S2", 52, synth.Camera(reversed_z=False),
                                                  dict(ao_format=1, f16_rounding=1, intensity=1.1), True),
    # ... and at the metric's frame size in the canonical rounding (3 h 42 min of interpreter time,
    # profiles/r06_reference_text_4k_f16_rtz_interpretation.log)
    "ref_s2_3840x2160_f16_rtz_checksums": (3840, 2160, "S2", 61, synth.DEFAULT_CAMERA, dict(ao_format=1), True),
}
# BASELINE config 2's size (1080p, the atrium frame the bench uses for it): an hour of interpreter time; the fixture keeps the result
# texture and a 64-bit order-sensitive checksum (tests.helpers.checksum) of each of the 17 buffers instead of the buffers (35 MB)
CHECKSUM_CASES = ("ref_s3_1920x1080_sponza_r8_checksums", "ref_s2_3840x2160_r8_checksums",
                  "ref_s2_1920x1080_f16_rtz_checksums", "ref_s2_1920x1080_f16_rtne_convz_checksums",
                  "ref_s2_3840x2160_f16_rtz_checksums")
# fixtures whose UNORM8 / f16 encode-decode is NOT the oracle's (VERDICT r4 weak #1a)
NUMPY_CODEC_CASES = ("ref_s2_150x86_r8_numpy_codecs", "ref_s2_134x70_f16_rtz_numpy_codecs")
# frames with NaN texels: compare bit patterns with any-NaN == any-NaN (tests.helpers.nan_aware_equal)
HOSTILE_CASES = ("ref_s2h_516x260_hostile_r8",)


AO_CS = "/root/reference/Assets/MiniEngineAO/AmbientOcclusion.cs"
SHADER_FILES = ("Downsample1", "Downsample2", "Render", "Upsample")

# shader property name -> key in the oracle's buffer dict
ORACLE_KEY = {"LinearDepth": "linear_depth", "AmbientOcclusion": "result"}
for _k in range(1, 5):
    ORACLE_KEY.update({f"LowDepth{_k}": f"low_depth{_k}", f"TiledDepth{_k}": f"tiled_depth{_k}",
                       f"Occlusion{_k}": f"occlusion{_k}", f"Combined{_k}": f"combined{_k}",
                       f"OcclusionHQ{_k}": f"occlusion_hq{_k}"})


def shader_sources():
    return {n: open(os.path.join(SHADERS, n + ".compute")).read() for n in SHADER_FILES}


def kernel_numthreads(src):
    """[numthreads] of every #pragma kernel variant, read from the shader text."""
    out = {}
    for name, text in src.items():
        out[name] = {}
        for kernel in HI.kernel_variants(text):
            prog, entry = HI.compile_kernel(text, kernel)
            m = HI.Machine(prog)
            out[name][kernel] = tuple(int(m.eval(d, [])[1][0]) for d in prog.funcs[entry].numthreads)
    return out


def record_reference_commands(s, want_interp=False):
    """Run the reference's own C# (AmbientOcclusion.cs: DoLazyInitialization + RebuildCommandBuffers)
    through oracle/csharp_interp.py against recording Unity mocks."""
    from tests.golden import unity_mocks as U
    src = shader_sources()
    props = {"_noiseFilterTolerance": s.noise_filter_tolerance, "_blurTolerance": s.blur_tolerance,
             "_upsampleTolerance": s.upsample_tolerance, "_thicknessModifier": s.thickness_modifier,
             "_intensity": s.intensity}
    stereo = bool(s.single_pass_stereo)
    assert not stereo or s.width % 2 == 0
    out = U.run_component(AO_CS, kernel_numthreads(src), width=s.width // 2 if stereo else s.width,
                          height=s.height, near=s.near_clip, far=s.far_clip, proj00=s.proj00,
                          reversed_z=s.reversed_z, properties=props, stereo=stereo, want_interp=True)
    cmd, result_rt, it, comp = out
    return (src, cmd, result_rt, it, comp) if want_interp else (src, cmd, result_rt)


def hq_render_commands(s, it, comp, level):
    """The reference's PushRenderCommands (AO.cs:660-747) with the NON-tiled LowDepth<level> as the
    source: yields the constant block of the !source.isTiled branch (AO.cs:679).  The method always
    picks main_interleaved; the kernel name and group counts are replaced by Render.main's."""
    from tests.golden import unity_mocks as U
    cls = comp.cls
    tan = it.call_method(comp, cls, "CalculateTanHalfFovHeight", [])
    cmd = U.CommandBuffer()
    it.call_method(comp, cls, "PushRenderCommands",
                   [cmd, comp.f[f"_lowDepth{level}"], comp.f[f"_occlusion{level}"], tan])
    d = cmd.dispatches[-1]
    w, h = O.level_dims(s.width, s.height, level)
    return {"shader": "Render", "kernel": "main", "groups": ((w + 15) // 16, (h + 15) // 16, 1),
            "tex": {"DepthTex": f"LowDepth{level}", "Occlusion": f"OcclusionHQ{level}"}, "const": d["const"]}


def numpy_codecs(f16_rounding):
    """Storage conversions restated from the format definitions with NumPy only -- no call into oracle/
    (VERDICT r4 weak #1a: the other fixtures' conversions are the oracle's own C functions).
    f16: IEEE binary16 via NumPy's own float16 (round-to-nearest-even, overflow -> inf); round-toward-
    zero = step the RTNE result one code toward zero whenever it rounded away (inf -> 65504, the clamp of
    DESIGN.md section 2 included).  UNORM8: n / 255 in binary32; store = trunc(saturate(x) * 255 + 0.5),
    one binary32 rounding per operation, NaN -> 0 (D3D11 float -> UNORM rule)."""
    def f16_decode(bits):
        return np.array([bits], np.uint16).view(np.float16).astype(np.float32)[0]

    def f16_encode(v):
        v32 = np.array([v], np.float32)
        with np.errstate(over="ignore"):
            h = v32.astype(np.float16)
        bits = int(h.view(np.uint16)[0])
        if f16_rounding == O.F16_RTZ and not np.isnan(v32[0]) and abs(np.float32(h[0])) > abs(v32[0]):
            bits -= 1                                   # sign-magnitude: one code toward zero
        return bits

    def r8_decode(n):
        return np.float32(int(n)) / np.float32(255)

    def r8_encode(v):
        v = np.float32(v)
        if np.isnan(v):
            return 0
        c = min(max(v, np.float32(0)), np.float32(1))
        t = np.float32(c * np.float32(255))
        return int(np.float32(t + np.float32(0.5)))
    return dict(decode=f16_decode, encode=f16_encode), dict(decode=r8_decode, encode=r8_encode)


def textures(s, cmd, result_rt, independent_codecs=False):
    """Render textures exactly as the reference allocates them (names, dims, slices, formats come
    from the recorded GetTemporaryRT(Array) calls and the persistent result RT).  The only
    deviation is this project's extension: AO targets are RHalf instead of R8 when s.ao_format
    is F16; the f32->f16 store rounding is the canonical choice of DESIGN.md section 2."""
    L = O.lib()
    f16 = dict(decode=lambda v: np.float32(L.meao_oracle_f16_to_f32(int(v))),
               encode=lambda v: L.meao_oracle_f32_to_f16(float(v), s.f16_rounding))
    r8 = dict(decode=lambda v: np.float32(L.meao_oracle_unorm8_to_f32(int(v))),
              encode=lambda v: L.meao_oracle_f32_to_unorm8(float(v)))
    if independent_codecs:
        f16, r8 = numpy_codecs(s.f16_rounding)
    allocs = dict(cmd.allocs)
    allocs["AmbientOcclusion"] = (result_rt.width, result_rt.height, 1, result_rt.format._name.split(".")[-1])
    for k in s.hq_level_list():                               # this project's extra targets
        allocs[f"OcclusionHQ{k}"] = allocs[f"Occlusion{k}"]
    arrs, tex = {}, {}
    for name, (w, h, slices, fmt) in allocs.items():
        if fmt == "R8" and s.ao_format == O.AO_F16:
            fmt = "RHalf"
        dt = {"RFloat": np.float32, "RHalf": np.uint16, "R8": np.uint8}[fmt]
        a = np.zeros((slices, h, w), dt)
        arrs[ORACLE_KEY[name]] = a if slices > 1 else a[0]
        tex[name] = HI.Texture(a, **({} if fmt == "RFloat" else (f16 if fmt == "RHalf" else r8)))
    return arrs, tex


def variant_dispatches(s, cmd, it, comp):
    """The recorded dispatch list, re-wired for the variants (see the module docstring)."""
    out = []
    hq = s.hq_level_list()
    exhaustive = s.sample_set == O.SAMPLES_EXHAUSTIVE
    for d in cmd.dispatches:
        d = dict(d, tex=dict(d["tex"]), const=dict(d["const"]), defines={})
        if d["shader"] == "Render":
            level = int(d["tex"]["Occlusion"][-1])
            renders = [d] + ([hq_render_commands(s, it, comp, level)] if level in hq else [])
            for r in renders:
                r.setdefault("defines", {})
                if exhaustive:                                # AO.cs:709 FIXME: the host never builds this table
                    consts = (O.render_constants_hq if r["kernel"] == "main" else O.render_constants)(s, level)
                    r["const"] = dict(r["const"], gSampleWeightTable=[np.float32(v) for v in consts.sample_weight])
                    r["defines"] = {"SAMPLE_EXHAUSTIVELY": "1"}
                out.append(r)
            continue
        if d["shader"] == "Upsample":
            low_level = int(d["tex"]["LoResDB"][-1])
            if low_level in hq:
                d["kernel"] = {"main": "main_premin", "main_blendout": "main_premin_blendout"}[d["kernel"]]
                d["tex"]["LoResAO2"] = f"OcclusionHQ{low_level}"
        out.append(d)
    return out


def run_reference_shaders(depth, s, log=print, independent_codecs=False):
    """The reference end to end: its C# decides allocations, bindings, constants and dispatch sizes,
    its HLSL does the arithmetic; both are interpreted from the source text under /root/reference."""
    assert s.depth_format == O.DEPTH_F32, "the reference runs on an RFloat depth"
    assert s.num_levels == 4, "the reference always runs 4 levels"
    src, cmd, result_rt, it, comp = record_reference_commands(s, want_interp=True)
    arrs, tex = textures(s, cmd, result_rt, independent_codecs)
    depth_tex = HI.Texture(np.ascontiguousarray(depth, np.float32)[None])
    rev = {"UNITY_REVERSED_Z": "1"} if s.reversed_z else {}
    t0 = time.time()
    for d in variant_dispatches(s, cmd, it, comp):
        text = src[d["shader"]]
        variant = dict(HI.kernel_variants(text)[d["kernel"]], **rev, **d.get("defines", {}))
        entry = variant.get("MAIN", d["kernel"])
        prog = HI.Parser(HI.lex(HI.preprocess(text, variant))).program()
        HI.attach_semantics(prog, entry, HI.preprocess(text, variant))
        m = HI.Machine(prog)
        for prop, what in d["tex"].items():
            if prop not in prog.resources:
                continue                                  # stale binding of another kernel variant
            if isinstance(what, str):
                m.bind[prop] = tex[what]
            elif type(what).__name__ == "RenderTexture":  # the persistent _result RenderTexture
                m.bind[prop] = tex["AmbientOcclusion"]
            else:                                         # BuiltinRenderTextureType.ResolvedDepth
                m.bind[prop] = depth_tex
        for name, (typ, length) in prog.uniforms.items():
            vals = d["const"][name]
            n = HI.parse_type(typ)[1]
            if length is None:
                m.const[name] = ("f", list(vals[:n]))
            else:
                m.const[name] = [("f", list(vals[i * n:(i + 1) * n])) for i in range(len(vals) // n)]
        m.dispatch(entry, d["groups"])
        log(f"  {d['shader']}.{d['kernel']} {d['groups']} done {time.time() - t0:.1f}s")
    return arrs, cmd


def make_depth(kind, w, h, seed, cam, sky):
    if kind == "S3":
        depth = synth.atrium(w, h, cam)
    elif kind == "S2H":                                                # NaN / inf / denormal / negative / > 1 texels
        depth = H.hostile_frame(w, h, seed, cam=cam, density=0.004)
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
        ref, _ = run_reference_shaders(depth, s, independent_codecs=name in NUMPY_CODEC_CASES)
        want = O.run(depth, s)
        same = (lambda a, b: H.nan_aware_equal(a, b)[0]) if name in HOSTILE_CASES else np.array_equal
        bad = [k for k in ref if not same(ref[k], want[k])]
        print("  interpreter vs oracle: %s" % ("all %d buffers identical" % len(ref) if not bad else "DIFFER: %s" % bad))
        if name in CHECKSUM_CASES:
            np.savez_compressed(os.path.join(HERE, name + ".npz"), result=ref["result"], depth_checksum=np.uint64(H.checksum(depth)),
                                **{"checksum_" + k: np.uint64(H.checksum(v)) for k, v in ref.items()})
        else:
            np.savez_compressed(os.path.join(HERE, name + ".npz"), depth=depth, **ref)


if __name__ == "__main__":
    main(sys.argv[1:])
