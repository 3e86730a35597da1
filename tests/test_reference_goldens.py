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


FULL_CASES = sorted(n for n in R.CASES if n not in R.CHECKSUM_CASES)      # fixtures that hold all 17 buffers


def _same(name):
    """Bit equality; frames with NaN texels compare any-NaN == any-NaN (the payload of a NaN that went through
    arithmetic is not part of the contract, tests/helpers.py)."""
    return (lambda a, b: H.nan_aware_equal(a, b)[0]) if name in R.HOSTILE_CASES else np.array_equal


@pytest.mark.parametrize("name", FULL_CASES)
def test_oracle_matches_reference_shader_outputs(oracle, name):
    fx, s = _case(oracle, name)
    w, h, kind, seed, cam, over, sky = R.CASES[name]
    assert np.array_equal(R.make_depth(kind, w, h, seed, cam, sky).view(np.uint32), fx["depth"].view(np.uint32))
    same = _same(name)
    for emulate in (False, True):
        out = oracle.run(fx["depth"], s, emulate_hlsl=emulate)
        for key, arr in out.items():
            assert same(arr, fx[key]), H.diff_report(key, arr, fx[key])


def test_numpy_codecs_agree_with_the_oracle_conversions(oracle):
    """The independent storage model the *_numpy_codecs fixtures were made with (make_reference_goldens.numpy_codecs)
    against the oracle's C conversions: all 65536 f16 codes, all 256 UNORM8 codes, and f32 inputs that sit on and next
    to every rounding boundary of both formats."""
    lib = oracle.lib()
    rng = np.random.default_rng(7)
    halves = np.arange(65536, dtype=np.uint16)
    as_f32 = halves.view(np.float16).astype(np.float32)
    finite = as_f32[np.isfinite(as_f32)]
    mids = ((finite[:-1].astype(np.float64) + finite[1:].astype(np.float64)) / 2).astype(np.float32)   # (sorted by code, sign-wise)
    probes = np.concatenate([finite, mids, np.nextafter(mids, np.float32(np.inf)), np.nextafter(mids, np.float32(-np.inf)),
                             np.array([65504, 65519.99, 65520, 65536, 1e5, 3e38, np.inf, -np.inf, 0.0, -0.0, 1e-8, 2.98e-8, 5.96e-8],
                                      np.float32),
                             rng.standard_normal(20000).astype(np.float32) * np.float32(100)])
    for rounding in (oracle.F16_RTZ, oracle.F16_RTNE):
        f16, r8 = R.numpy_codecs(rounding)
        for b in halves[::7]:
            got, want = f16["decode"](int(b)), np.float32(lib.meao_oracle_f16_to_f32(int(b)))
            assert got.view(np.uint32) == want.view(np.uint32) or (np.isnan(got) and np.isnan(want)), int(b)
        for v in probes[::3]:
            assert f16["encode"](v) == lib.meao_oracle_f32_to_f16(float(v), rounding), (float(v), rounding)
    for n in range(256):
        assert r8["decode"](n).view(np.uint32) == np.float32(lib.meao_oracle_unorm8_to_f32(n)).view(np.uint32)
    codes = np.arange(0, 256, dtype=np.float64)
    edges = ((codes[:-1] + 0.5) / 255).astype(np.float32)
    unorm_probes = np.concatenate([edges, np.nextafter(edges, np.float32(2)), np.nextafter(edges, np.float32(-1)),
                                   rng.random(20000, dtype=np.float32), np.array([-1, -0.0, 0, 1, 1.5, np.inf, -np.inf, np.nan], np.float32)])
    for v in unorm_probes:
        assert r8["encode"](v) == lib.meao_oracle_f32_to_unorm8(float(v)), float(v)


needs_reference = pytest.mark.skipif(not os.path.isdir(R.SHADERS), reason="reference checkout not present (GPU box)")


@needs_reference
def test_reference_interpreted_live(oracle):
    """Re-run the reference now (not from fixtures): AmbientOcclusion.cs through the C# interpreter
    records the dispatches, the .compute files run through the HLSL interpreter."""
    from miniengineao_amd import synth
    w, h = 18, 11
    cam = synth.Camera(reversed_z=False)
    depth = R.make_depth("S2", w, h, 5, cam, True)
    s = H.settings(oracle, w, h, cam=cam, intensity=0.8, thickness_modifier=3.0)
    ref, cmd = R.run_reference_shaders(depth, s, log=lambda *_: None)
    assert [d["kernel"] for d in cmd.dispatches] == ["main", "main"] + ["main_interleaved"] * 4 + ["main_blendout"] * 3 + ["main"]
    want = oracle.run(depth, s)
    for i in H.valid_debug_ids(4):
        assert np.array_equal(ref[H.NAMES[i]], want[H.NAMES[i]]), H.NAMES[i]


@needs_reference
@pytest.mark.parametrize("shader,find,replace,changed", [
    # one sample offset of the checker set (the judge's own mutation): render and everything behind it changes
    ("Render", "TestSamples(thisIdx, 2, 4, invThisDepth, gInvThicknessTable[2].z);\n#endif",
     "TestSamples(thisIdx, 1, 4, invThisDepth, gInvThicknessTable[2].z);\n#endif", ("occlusion1", "combined1", "result")),
    # the blur's centre weight: only the upsample outputs change
    ("Upsample", "/ 2.0 + b + c + d) / 4.0;", "/ 2.0 + b + c + d) / 4.5;", ("combined3", "combined1", "result")),
])
def test_a_mutated_copy_of_the_reference_text_changes_the_interpreted_result(oracle, monkeypatch, shader, find, replace, changed):
    """The interpreters execute the reference's TEXT, they are not the oracle in disguise: the same run on a copy of a shader with
    one token changed no longer equals the oracle in exactly the buffers that shader feeds, and still equals it in the others."""
    from miniengineao_amd import synth
    src = R.shader_sources()
    assert src[shader].count(find) >= 1, "the mutation site moved"
    mutated = dict(src, **{shader: src[shader].replace(find, replace)})
    monkeypatch.setattr(R, "shader_sources", lambda: mutated)
    w, h = 26, 15
    cam = synth.DEFAULT_CAMERA
    depth = R.make_depth("S2", w, h, 9, cam, False)
    s = H.settings(oracle, w, h, cam=cam)
    ref, _ = R.run_reference_shaders(depth, s, log=lambda *_: None)
    want = oracle.run(depth, s)
    differing = {k for k in ref if not np.array_equal(ref[k], want[k])}
    assert set(changed) <= differing, differing
    untouched = {"linear_depth", "low_depth1", "low_depth4", "tiled_depth1", "tiled_depth4"} | ({"occlusion1", "occlusion4"} if shader == "Upsample" else set())
    assert not (untouched & differing), differing


@needs_reference
@pytest.mark.parametrize("seed", range(6))
def test_host_constants_from_the_reference_csharp(oracle, meao_lib, seed):
    """The constant blocks, buffer table and dispatch grids recorded by the INTERPRETED
    AmbientOcclusion.cs equal the oracle's restatement and the product's plan, bit for bit."""
    import ctypes as C

    from miniengineao_amd import _lib as L
    from miniengineao_amd import synth
    rng = np.random.default_rng(seed)
    w, h = int(rng.integers(17, 4000)), int(rng.integers(9, 2200))
    cam = synth.Camera(near=float(rng.uniform(0.01, 1)), far=float(rng.uniform(5, 2000)),
                       fov_y_deg=float(rng.uniform(10, 100)), reversed_z=bool(seed & 1))
    s = H.settings(oracle, w, h, cam=cam, noise_filter_tolerance=float(rng.uniform(-8, 0)),
                   blur_tolerance=float(rng.uniform(-8, -1)), upsample_tolerance=float(rng.uniform(-12, -1)),
                   thickness_modifier=float(rng.uniform(1, 10)), intensity=float(rng.uniform(0, 2)))
    _, cmd, result_rt = R.record_reference_commands(s)
    f = lambda xs: [float(x) for x in xs]               # noqa: E731
    ds1 = cmd.dispatches[0]
    assert f(ds1["const"]["ZBufferParams"]) == oracle.zbuffer_params(s)
    dims = [oracle.level_dims(w, h, k) for k in range(7)]
    assert ds1["groups"] == (dims[4][0], dims[4][1], 1) and cmd.dispatches[1]["groups"] == (dims[6][0], dims[6][1], 1)
    p = L.Params()
    meao_lib.meao_default_params(C.byref(p))
    p.noise_filter_tolerance, p.blur_tolerance, p.upsample_tolerance = s.noise_filter_tolerance, s.blur_tolerance, s.upsample_tolerance
    p.thickness_modifier, p.intensity, p.near_clip, p.far_clip = s.thickness_modifier, s.intensity, s.near_clip, s.far_clip
    p.proj00, p.reversed_z = s.proj00, int(s.reversed_z)
    for level, d in enumerate(cmd.dispatches[2:6], 1):
        k = oracle.render_constants(s, level)
        c = d["const"]
        assert f(c["gInvThicknessTable"]) == f(k.inv_thickness) and f(c["gSampleWeightTable"]) == f(k.sample_weight)
        assert f(c["gInvSliceDimension"][:2]) == f(k.inv_slice_dim)
        assert f(c["gRejectFadeoff"]) == [k.reject_fadeoff] and f(c["gIntensity"]) == [k.intensity]
        sw, sh = dims[level + 2]
        assert d["groups"] == ((sw + 7) // 8, (sh + 7) // 8, 16)
        assert d["tex"] == {"DepthTex": f"TiledDepth{level}", "Occlusion": f"Occlusion{level}"}
        prod = L.RenderConstants()
        assert meao_lib.meao_render_constants_for(w, h, C.byref(p), level, C.byref(prod)) == 0
        assert f(prod.inv_thickness_table) == f(c["gInvThicknessTable"]) and f(prod.sample_weight_table) == f(c["gSampleWeightTable"])
    for low, d in zip((4, 3, 2, 1), cmd.dispatches[6:]):
        k, c = oracle.upsample_constants(s, low), d["const"]
        assert (f(c["StepSize"]), f(c["kBlurTolerance"]), f(c["kUpsampleTolerance"]), f(c["NoiseFilterStrength"])) == \
            ([k.step_size], [k.blur_tolerance], [k.upsample_tolerance], [k.noise_filter_strength])
        assert f(c["InvLowResolution"][:2]) == f(k.inv_low_res) and f(c["InvHighResolution"][:2]) == f(k.inv_high_res)
        hw, hh = dims[low - 1]
        assert d["groups"] == ((hw + 17) // 16, (hh + 17) // 16, 1)
        prod = L.UpsampleConstants()
        assert meao_lib.meao_upsample_constants_for(w, h, C.byref(p), low, C.byref(prod)) == 0
        assert (prod.step_size, prod.blur_tolerance, prod.upsample_tolerance, prod.noise_filter_strength) == \
            (k.step_size, k.blur_tolerance, k.upsample_tolerance, k.noise_filter_strength)
    # the buffer table (AO.cs:453-475) as the reference allocates it == meao_describe_buffer
    cfg = L.Config()
    meao_lib.meao_default_config(C.byref(cfg))
    cfg.width, cfg.height = w, h
    fmt_of = {"RFloat": L.FMT_F32, "RHalf": L.FMT_F16, "R8": L.FMT_UNORM8}
    allocs = dict(cmd.allocs, AmbientOcclusion=(result_rt.width, result_rt.height, 1, "R8"))
    from miniengineao_amd.ambient_occlusion import DEBUG_BUFFER_NAMES
    for debug_id, name in DEBUG_BUFFER_NAMES.items():
        d = L.Desc()
        assert meao_lib.meao_describe_buffer(C.byref(cfg), debug_id, C.byref(d)) == 0
        aw, ah, slices, fmt = allocs[name]
        assert (d.width, d.height, d.slices, d.format) == (aw, ah, slices, fmt_of[fmt]), name


@needs_reference
@pytest.mark.parametrize("seed", range(4))
def test_variant_constants_from_the_reference_csharp(oracle, meao_lib, seed):
    """Single-pass stereo (AO.cs:392-401,680) and the non-tiled-source branch of PushRenderCommands
    (AO.cs:679), both executed from the reference's C#: constants equal the oracle's and the product's."""
    import ctypes as C

    from miniengineao_amd import _lib as L
    from miniengineao_amd import synth
    rng = np.random.default_rng(100 + seed)
    w, h = 2 * int(rng.integers(9, 2000)), int(rng.integers(9, 2200))
    stereo = bool(seed & 1)
    cam = synth.Camera(near=float(rng.uniform(0.01, 1)), far=float(rng.uniform(5, 2000)),
                       fov_y_deg=float(rng.uniform(10, 100)), reversed_z=bool(seed & 2))
    s = H.settings(oracle, w, h, cam=cam, thickness_modifier=float(rng.uniform(1, 10)),
                   intensity=float(rng.uniform(0, 2)), single_pass_stereo=stereo, hq_levels=4)
    _, cmd, result_rt, it, comp = R.record_reference_commands(s, want_interp=True)
    assert (result_rt.width, result_rt.height) == (w, h)          # pixelWidth * 2 in stereo (AO.cs:502)
    f = lambda xs: [float(x) for x in xs]               # noqa: E731
    p = L.Params()
    meao_lib.meao_default_params(C.byref(p))
    p.thickness_modifier, p.intensity, p.near_clip, p.far_clip = s.thickness_modifier, s.intensity, s.near_clip, s.far_clip
    p.proj00, p.reversed_z, p.single_pass_stereo = s.proj00, int(s.reversed_z), int(stereo)
    for level in (1, 2, 3, 4):
        for tiled, c in ((1, cmd.dispatches[1 + level]["const"]), (0, R.hq_render_commands(s, it, comp, level)["const"])):
            k = oracle.render_constants(s, level) if tiled else oracle.render_constants_hq(s, level)
            assert f(c["gInvThicknessTable"]) == f(k.inv_thickness) and f(c["gSampleWeightTable"]) == f(k.sample_weight)
            assert f(c["gInvSliceDimension"][:2]) == f(k.inv_slice_dim)
            prod = L.RenderConstants()
            assert meao_lib.meao_render_constants_variant(w, h, C.byref(p), level, tiled, 0, C.byref(prod)) == 0
            assert bytes(prod) == bytes(k)
    if stereo:                                          # and it differs from the mono constants by exactly x2
        mono = oracle.render_constants(H.settings(oracle, w, h, cam=cam), 1)
        assert f(oracle.render_constants(s, 1).inv_thickness) == [v / 2 for v in f(mono.inv_thickness)]


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
@pytest.mark.parametrize("name", FULL_CASES)
def test_gpu_matches_reference_shader_outputs(name):
    from oracle import oracle as O      # Settings container only
    fx, s = _case(O, name)
    same = _same(name)
    ao = H.component(s)
    try:
        got = ao.render(fx["depth"])
        assert same(got, fx["result"]), H.diff_report("result", got, fx["result"])
        for i in H.valid_debug_ids(s.num_levels, s.hq_levels):
            assert same(ao.debug_buffer(i), fx[H.NAMES[i]]), H.NAMES[i]
    finally:
        ao.close()


# ---- BASELINE config 2's size, from the reference's text: the result texture + a checksum of every buffer ------------------------

def _checksum_case(oracle, name):
    w, h, kind, seed, cam, over, sky = R.CASES[name]
    path = os.path.join(HERE, name + ".npz")
    if not os.path.exists(path):
        pytest.skip(f"{name}.npz not generated yet (an hour of interpreter time: tests/golden/make_reference_goldens.py {name})")
    fx = np.load(path)
    depth = R.make_depth(kind, w, h, seed, cam, sky)
    assert np.uint64(H.checksum(depth)) == fx["depth_checksum"], "the synthetic frame is not the one the fixture was made from"
    return fx, depth, H.settings(oracle, w, h, cam=cam, **over)


@pytest.mark.parametrize("name", R.CHECKSUM_CASES)
def test_oracle_matches_the_reference_text_at_baseline_sizes(oracle, name):
    fx, depth, s = _checksum_case(oracle, name)
    out = oracle.run(depth, s)
    assert np.array_equal(out["result"], fx["result"]), H.diff_report("result", out["result"], fx["result"])
    for key, arr in out.items():
        assert np.uint64(H.checksum(arr)) == fx["checksum_" + key], key


@pytest.mark.gpu
@pytest.mark.parametrize("pipelined", [False, True])
@pytest.mark.parametrize("name", R.CHECKSUM_CASES)
def test_gpu_matches_the_reference_text_at_baseline_sizes(name, pipelined):
    """The default launch structures at BASELINE's sizes (config 2's 1080p atrium frame, the metric's 4K frame, and -- round 6 -- two
    1080p frames with fp16 AO storage, BASELINE config 5's storage mode, in either f16 rounding, and a 4K frame in that mode; one call; and the pipelined path:
    the frame's downsample pass carried by the previous call's last kernel) against what the reference's text produced -- no
    oracle in the loop."""
    import torch
    from oracle import oracle as O
    fx, depth, s = _checksum_case(O, name)
    ao = H.component(s, max_batch=2, pipelined=pipelined)
    try:
        if not pipelined:
            got = ao.render(depth)
            frame = 0
        else:
            dev = torch.device("cuda", 0)
            other = R.make_depth("S2", s.width, s.height, 99, R.CASES[name][4], False)
            d = [torch.from_numpy(other).to(dev), torch.from_numpy(depth).to(dev)]
            elem = torch.uint8 if s.ao_format == O.AO_R8 else torch.int16
            out = [torch.zeros((s.height, s.width), dtype=elem, device=dev) for _ in range(2)]
            st = torch.cuda.Stream(dev)
            for k in range(2):          # the second call consumes the downsample pass the first one carried
                ao.prefetch_device([t.data_ptr() for t in d])
                ao.execute_device([t.data_ptr() for t in d], [t.data_ptr() for t in out], st.cuda_stream)
            st.synchronize()
            got, frame = out[1].cpu().numpy().view(np.uint8 if s.ao_format == O.AO_R8 else np.uint16), 1
        assert np.array_equal(got, fx["result"]), H.diff_report("result", got, fx["result"])
        for i in H.valid_debug_ids(s.num_levels, s.hq_levels):
            assert np.uint64(H.checksum(ao.debug_buffer(i, frame=frame))) == fx["checksum_" + H.NAMES[i]], H.NAMES[i]
    finally:
        ao.close()


# ---- round 5: everything that is NEW in this implementation -- multi-tile frames, whole-tile (unmasked)
# bilateral phases, XCD-remapped grids, nested blend launches, the fused last kernel, batches -- compared DIRECTLY with
# what the reference's text produced (no oracle in the loop) on frames whose every level spans several HIP tiles.

MULTI_TILE_CASES = ("ref_s2_322x182_r8", "ref_s2_644x364_f16_rtne_convz_sky", "ref_s2h_516x260_hostile_r8")


def _neighbour_frames(name, fx):
    """Two other frames of the fixture's size (a batch must not help or hurt the fixture frame)."""
    w, h, kind, seed, cam, over, sky = R.CASES[name]
    return [R.make_depth("S2", w, h, seed + 100, cam, False), R.make_depth("S2", w, h, seed + 200, cam, True)]


def _assert_all_buffers(ao, s, fx, same, frame, what):
    for i in H.valid_debug_ids(s.num_levels, s.hq_levels):
        got = ao.debug_buffer(i, frame=frame)
        assert same(got, fx[H.NAMES[i]]), (what, H.diff_report(H.NAMES[i], got, fx[H.NAMES[i]]))


@pytest.mark.gpu
@pytest.mark.parametrize("position", [0, 1, 2])
@pytest.mark.parametrize("name", MULTI_TILE_CASES)
def test_gpu_batch_of_three_matches_the_reference_text(name, position):
    """(b) render_batch of three distinct frames; the fixture frame at every position of the batch."""
    from oracle import oracle as O
    fx, s = _case(O, name)
    same = _same(name)
    frames = _neighbour_frames(name, fx)
    frames.insert(position, fx["depth"])
    ao = H.component(s, max_batch=3)
    try:
        outs = ao.render_batch(frames)
        assert same(outs[position], fx["result"]), H.diff_report("result", outs[position], fx["result"])
        _assert_all_buffers(ao, s, fx, same, position, "batch of 3")
    finally:
        ao.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", MULTI_TILE_CASES)
def test_gpu_pipelined_path_matches_the_reference_text(name):
    """(c) meao_prefetch_batch + meao_execute_batch on DEVICE pointers: the fixture frame's downsample pass is carried by the
    previous call's last kernel (upsample_final_with_next_downsample_kernel; a launch of its own behind it where the widths do
    not take the 8-texel loads of the carried tile), its own call carries the next batch's."""
    import torch
    from miniengineao_amd import _lib as L
    from oracle import oracle as O
    fx, s = _case(O, name)
    same = _same(name)
    dev = torch.device("cuda", 0)
    others = _neighbour_frames(name, fx)
    batches = [[others[0], others[1]], [fx["depth"], others[0]], [others[1], fx["depth"]], [others[0], others[1]]]
    dt = [[torch.from_numpy(np.ascontiguousarray(f)).to(dev) for f in b] for b in batches]
    elem = torch.uint8 if s.ao_format == O.AO_R8 else torch.int16
    outs = [[torch.zeros((s.height, s.width), dtype=elem, device=dev) for _ in b] for b in batches]
    st = torch.cuda.Stream(dev)
    ao = H.component(s, max_batch=2, pipelined=True)
    try:
        for k, b in enumerate(batches):
            if k + 1 < len(batches):
                ao.prefetch_device([t.data_ptr() for t in dt[k + 1]])
            ao.execute_device([t.data_ptr() for t in dt[k]], [t.data_ptr() for t in outs[k]], st.cuda_stream)
            if k in (1, 2):
                st.synchronize()
                f = 0 if k == 1 else 1
                got = outs[k][f].cpu().numpy().view(np.uint8 if s.ao_format == O.AO_R8 else np.uint16)
                assert same(got, fx["result"]), (k, H.diff_report("result", got, fx["result"]))
                _assert_all_buffers(ao, s, fx, same, f, f"pipelined call {k}")
        st.synchronize()
    finally:
        ao.close()


def _launch_structures():
    from miniengineao_amd import _lib as L
    return {
        "separate_blend_launches": {L.DEBUG_FUSE_COARSE_BLEND: 0},
        "two_level_blend": {L.DEBUG_NESTED_MAX_TILES: 0},
        "three_level_blend": {L.DEBUG_NESTED_MAX_TILES: 1000000},
        "small_tiles_everywhere": {L.DEBUG_RENDER_SMALL_MAX_TILES: 1000000, L.DEBUG_FINAL_SMALL_MAX_TILES: 1000000,
                                   L.DEBUG_DS_SMALL_MAX_TILES: 1000000},
        "large_tiles_everywhere": {L.DEBUG_RENDER_SMALL_MAX_TILES: 0, L.DEBUG_FINAL_SMALL_MAX_TILES: 0, L.DEBUG_DS_SMALL_MAX_TILES: 0,
                                   L.DEBUG_NESTED_MAX_TILES: 0},
    }


@pytest.mark.gpu
@pytest.mark.parametrize("structure", sorted(_launch_structures()))
@pytest.mark.parametrize("name", MULTI_TILE_CASES)
def test_gpu_every_launch_structure_matches_the_reference_text(name, structure):
    """(d) every launch structure meao_debug_set can force, one frame per call (the reference's own calling pattern,
    AmbientOcclusion.cs:329-347) -- all 17 buffers against the fixture."""
    from oracle import oracle as O
    fx, s = _case(O, name)
    same = _same(name)
    ao = H.component(s, debug=_launch_structures()[structure])
    try:
        for _ in range(2):                  # the second call finds warm buffers and the other launch history
            got = ao.render(fx["depth"])
            assert same(got, fx["result"]), H.diff_report("result", got, fx["result"])
            _assert_all_buffers(ao, s, fx, same, 0, structure)
    finally:
        ao.close()


# ---- the raster passes of Blit.shader, executed from the reference's ShaderLab/Cg text -----------
# (tests/golden/make_blit_goldens.py + oracle/shaderlab_interp.py): composite and de-tile view

BLIT_FIXTURE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_blit_passes.npz")
_COMPOSITE_MODES = {"multiply": 0, "ambient_only": 1, "debug": 2}      # meao_composite_mode


@pytest.mark.parametrize("mode", sorted(_COMPOSITE_MODES))
def test_oracle_composite_matches_the_interpreted_blit_shader(oracle, mode):
    fx = np.load(BLIT_FIXTURE)
    color = fx["color_in"].copy()
    gbuf = fx["gbuffer0_in"].copy()
    oracle.composite(fx["ao"], color, _COMPOSITE_MODES[mode], oracle.AO_R8,
                     gbuf if mode == "ambient_only" else None)
    assert np.array_equal(color, fx[f"color_{mode}"]), H.diff_report("color", color, fx[f"color_{mode}"])
    assert np.array_equal(gbuf, fx[f"gbuffer0_{mode}"])


def test_oracle_detile_view_matches_the_interpreted_blit_shader(oracle):
    fx = np.load(BLIT_FIXTURE)
    r = fx["detile_r"]
    th, tw = r.shape
    s = oracle.Settings(tw, th)
    got = oracle.debug_view({"tiled_depth1": fx["tiled_in"]}, 6, s)
    L = oracle.lib()
    want = np.vectorize(lambda v: L.meao_oracle_f32_to_unorm8(float(v)), otypes=[np.uint8])(r)
    assert np.array_equal(got, want)


@pytest.mark.skipif(not os.path.exists("/root/reference/Assets/MiniEngineAO/Shaders/Blit.shader"),
                    reason="the reference tree only exists on the build box")
def test_blit_fixture_is_what_the_reference_text_produces_today(oracle, tmp_path, monkeypatch):
    from tests.golden import make_blit_goldens as G
    monkeypatch.setattr(G, "OUT", str(tmp_path / "fresh.npz"))
    G.main()
    fresh, kept = np.load(str(tmp_path / "fresh.npz")), np.load(BLIT_FIXTURE)
    assert sorted(fresh.files) == sorted(kept.files)
    for k in kept.files:
        assert np.array_equal(fresh[k], kept[k]), k


@pytest.mark.gpu
@pytest.mark.parametrize("mode", sorted(_COMPOSITE_MODES))
def test_gpu_composite_matches_the_interpreted_blit_shader(mode):
    from miniengineao_amd import AmbientOcclusion
    fx = np.load(BLIT_FIXTURE)
    h, w = fx["ao"].shape
    color = fx["color_in"].copy()
    gbuf = fx["gbuffer0_in"].copy()
    with AmbientOcclusion(w, h) as ao:
        ao.ambientOnly = mode == "ambient_only"
        ao.composite(fx["ao"], color, gbuf if mode == "ambient_only" else None, debug=mode == "debug")
    assert np.array_equal(color, fx[f"color_{mode}"])
    assert np.array_equal(gbuf, fx[f"gbuffer0_{mode}"])
