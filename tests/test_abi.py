"""C-ABI library: loads, exports every symbol include/meao.h declares, host-side plan functions
agree with the oracle, error behaviour.  No compute calls here (CPU suite, no GPU needed)."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from miniengineao_amd import _lib as L
from miniengineao_amd import synth
from tests import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
with open(os.path.join(ROOT, "include", "meao.h")) as _f:
    HEADER = _f.read()


def header_functions():
    return re.findall(r"MEAO_API\s+[\w\s\*]+?\b(meao_\w+)\s*\(", HEADER)


def test_header_declares_what_the_binding_binds():
    declared = set(header_functions())
    assert len(declared) >= 24
    assert declared == set(L.SIGNATURES), declared ^ set(L.SIGNATURES)


def test_library_exports_every_declared_symbol(meao_lib):
    out = subprocess.run(["nm", "-D", "--defined-only", L.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (meao_\w+)", out))
    assert set(header_functions()) <= exported
    assert all(not s.startswith("meao_oracle") for s in exported), "oracle code must not be in the product library"
    assert meao_lib.meao_abi_version() == L.ABI_VERSION


def test_product_library_does_not_link_the_oracle():
    out = subprocess.run(["objdump", "-p", L.LIB_PATH], capture_output=True, text=True, check=True).stdout
    needed = re.findall(r"NEEDED\s+(\S+)", out)
    assert any("amdhip64" in n for n in needed)
    assert not any("oracle" in n or "torch" in n or "python" in n for n in needed), needed


def test_struct_layouts_match_header():
    # every member is a 4-byte scalar except meao_desc.bytes (u64, naturally aligned)
    assert C.sizeof(L.Config) == 12 * 4 and C.sizeof(L.Params) == 11 * 4
    assert C.sizeof(L.Desc) == 32 and L.Desc.bytes.offset == 24
    assert C.sizeof(L.RenderConstants) == 28 * 4 and C.sizeof(L.UpsampleConstants) == 8 * 4
    for struct, cname in ((L.Config, "meao_config"), (L.Params, "meao_params"), (L.Desc, "meao_desc")):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (cname, cname), HEADER, re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        names = []
        for decl in body.split(";"):
            decl = decl.strip()
            if decl:
                names += [n.strip() for n in decl.split(None, 1)[1].split(",")]
        assert names == [f[0] for f in struct._fields_], cname


def test_defaults_are_the_reference_defaults(meao_lib):
    p = L.Params()
    meao_lib.meao_default_params(C.byref(p))
    # AmbientOcclusion.cs:20-52
    assert (p.noise_filter_tolerance, p.upsample_tolerance, p.thickness_modifier, p.intensity) == (0.0, -12.0, 1.0, 1.0)
    assert p.blur_tolerance == np.float32(-4.6) and p.struct_size == C.sizeof(L.Params)
    c = L.Config()
    meao_lib.meao_default_config(C.byref(c))
    assert (c.num_levels, c.ao_format, c.f16_rounding, c.max_batch) == (4, L.AO_R8, L.F16_RTZ_CLAMP, 1)


def _params(meao_lib, s):
    p = L.Params()
    meao_lib.meao_default_params(C.byref(p))
    p.noise_filter_tolerance, p.blur_tolerance = s.noise_filter_tolerance, s.blur_tolerance
    p.upsample_tolerance, p.thickness_modifier, p.intensity = s.upsample_tolerance, s.thickness_modifier, s.intensity
    p.near_clip, p.far_clip, p.proj00, p.reversed_z = s.near_clip, s.far_clip, s.proj00, int(s.reversed_z)
    p.single_pass_stereo = int(s.single_pass_stereo)
    return p


@pytest.mark.parametrize("seed", range(12))
def test_plan_constants_match_oracle_bitwise(meao_lib, oracle, seed):
    """Two independent implementations (C oracle, C++ product) of AO.cs:561-573,660-771."""
    rng = np.random.default_rng(seed)
    w, h = int(rng.integers(1, 8000)), int(rng.integers(1, 4400))
    s = H.settings(oracle, w, h, cam=synth.Camera(near=float(rng.uniform(0.01, 1)), far=float(rng.uniform(5, 2000)),
                                                   fov_y_deg=float(rng.uniform(10, 100)), reversed_z=bool(seed & 1)),
                   noise_filter_tolerance=float(rng.uniform(-8, 0)), blur_tolerance=float(rng.uniform(-8, -1)),
                   upsample_tolerance=float(rng.uniform(-12, -1)), thickness_modifier=float(rng.uniform(1, 10)),
                   intensity=float(rng.uniform(0, 2)), single_pass_stereo=bool(seed & 2),
                   sample_set=(seed >> 2) & 1)
    p = _params(meao_lib, s)
    zp = (C.c_float * 4)()
    assert meao_lib.meao_zbuffer_params(C.byref(p), C.byref(zp)) == 0
    assert list(zp) == oracle.zbuffer_params(s)
    for level in (1, 2, 3, 4):
        a, b = oracle.render_constants(s, level), L.RenderConstants()
        assert meao_lib.meao_render_constants_variant(w, h, C.byref(p), level, 1, s.sample_set, C.byref(b)) == 0
        assert bytes(a) == bytes(b)
        if s.sample_set == 0:
            assert meao_lib.meao_render_constants_for(w, h, C.byref(p), level, C.byref(b)) == 0
            assert bytes(a) == bytes(b)
        a = oracle.render_constants_hq(s, level)           # Render.main on the non-tiled LowDepth<level>
        assert meao_lib.meao_render_constants_variant(w, h, C.byref(p), level, 0, s.sample_set, C.byref(b)) == 0
        assert bytes(a) == bytes(b)
        a, b = oracle.upsample_constants(s, level), L.UpsampleConstants()
        assert meao_lib.meao_upsample_constants_for(w, h, C.byref(p), level, C.byref(b)) == 0
        assert bytes(a) == bytes(b)
    for level in range(7):
        ow, oh = C.c_int32(), C.c_int32()
        assert meao_lib.meao_level_dims(w, h, level, C.byref(ow), C.byref(oh)) == 0
        assert (ow.value, oh.value) == oracle.level_dims(w, h, level) == (-(-w // 2 ** level), -(-h // 2 ** level))


def test_buffer_table(meao_lib, oracle):
    """The 17 debug-visible buffers of AO.cs:453-475 / 789-808: dims, format, slices."""
    cfg = L.Config()
    meao_lib.meao_default_config(C.byref(cfg))
    cfg.width, cfg.height = 1920, 1080
    arrs = oracle.allocate(oracle.Settings(1920, 1080))
    total = 1920 * 1080 * 4     # + the input depth copy
    for i in range(1, 18):
        d = L.Desc()
        assert meao_lib.meao_describe_buffer(C.byref(cfg), i, C.byref(d)) == 0
        a = arrs[H.NAMES[i]]
        assert (d.slices, d.height, d.width) == ((16,) + a.shape[1:] if a.ndim == 3 else (1,) + a.shape)
        assert d.bytes == a.nbytes
        total += d.bytes
    assert round(total / 1e6, 1) == 20.0            # 20.0 MB at 1080p
    assert meao_lib.meao_describe_buffer(C.byref(cfg), 0, C.byref(d)) == L.ERR_INVALID_ARGUMENT
    for i in range(18, 22):                          # OcclusionHQ1..4: same table rows as Occlusion1..4
        assert meao_lib.meao_describe_buffer(C.byref(cfg), i, C.byref(d)) == 0
        a = arrs[H.NAMES[i - 8]]
        assert (d.slices, d.height, d.width, d.bytes) == (1,) + a.shape + (a.nbytes,)
    assert meao_lib.meao_describe_buffer(C.byref(cfg), 22, C.byref(d)) == L.ERR_INVALID_ARGUMENT


@pytest.mark.parametrize("w,h,fmt,total_mb,ren_ups_mb", [
    (1920, 1080, 0, 32.52, 15.81), (3840, 2160, 0, 130.06, 63.25), (3840, 2160, 1, 149.30, 82.49),
    (7680, 4320, 0, 520.22, 252.98), (7680, 4320, 1, 597.20, 329.96)])
def test_algorithmic_bytes_match_baseline_md(meao_lib, w, h, fmt, total_mb, ren_ups_mb):
    cfg = L.Config()
    meao_lib.meao_default_config(C.byref(cfg))
    cfg.width, cfg.height, cfg.ao_format = w, h, fmt
    b = (C.c_uint64 * L.NUM_PASSES)()
    assert meao_lib.meao_algorithmic_bytes(C.byref(cfg), C.byref(b)) == 0
    assert round(sum(b) / 1e6, 2) == total_mb
    assert round(sum(list(b)[1:]) / 1e6, 2) == ren_ups_mb
    assert b[L.PASS_NAMES.index("render_hq")] == 0
    cfg.hq_levels = 4                                 # + Render.main per level and LoResAO2 per upsample
    base = list(b)
    assert meao_lib.meao_algorithmic_bytes(C.byref(cfg), C.byref(b)) == 0
    a = 1 if fmt == 0 else 2
    px = [(-(-w // 2 ** k)) * (-(-h // 2 ** k)) for k in range(5)]
    assert b[L.PASS_NAMES.index("render_hq")] == sum((4 + a) * px[k] for k in range(1, 5))
    assert sum(b) - sum(base) == sum((4 + 2 * a) * px[k] for k in range(1, 5))


def test_argument_validation(meao_lib):
    ctx = C.c_void_p()
    cfg = L.Config()
    meao_lib.meao_default_config(C.byref(cfg))
    assert meao_lib.meao_create(None, C.byref(ctx)) == L.ERR_INVALID_ARGUMENT
    bad = L.Config.from_buffer_copy(cfg); bad.num_levels = 5
    assert meao_lib.meao_create(C.byref(bad), C.byref(ctx)) == L.ERR_INVALID_ARGUMENT
    assert b"num_levels" in meao_lib.meao_last_error(None)
    bad = L.Config.from_buffer_copy(cfg); bad.struct_size = 12
    assert meao_lib.meao_create(C.byref(bad), C.byref(ctx)) == L.ERR_INVALID_ARGUMENT
    bad = L.Config.from_buffer_copy(cfg); bad.max_batch = 65
    assert meao_lib.meao_create(C.byref(bad), C.byref(ctx)) == L.ERR_INVALID_ARGUMENT
    bad = L.Config.from_buffer_copy(cfg); bad.hq_levels = 5
    assert meao_lib.meao_create(C.byref(bad), C.byref(ctx)) == L.ERR_INVALID_ARGUMENT
    bad = L.Config.from_buffer_copy(cfg); bad.num_levels = 2; bad.hq_levels = 3
    assert meao_lib.meao_create(C.byref(bad), C.byref(ctx)) == L.ERR_INVALID_ARGUMENT
    bad = L.Config.from_buffer_copy(cfg); bad.sample_set = 2
    assert meao_lib.meao_create(C.byref(bad), C.byref(ctx)) == L.ERR_INVALID_ARGUMENT
    bad = L.Config.from_buffer_copy(cfg); bad.pipelined = 2
    assert meao_lib.meao_create(C.byref(bad), C.byref(ctx)) == L.ERR_INVALID_ARGUMENT
    assert meao_lib.meao_destroy(None) == 0
    assert meao_lib.meao_execute(None, None, 0, None, 0, None) == L.ERR_INVALID_ARGUMENT
    assert meao_lib.meao_prefetch_batch(None, 1, None) == L.ERR_INVALID_ARGUMENT
    assert meao_lib.meao_status_string(L.ERR_NO_DEVICE).startswith(b"no gfx950 device")
    p = L.Params()
    meao_lib.meao_default_params(C.byref(p))
    p.near_clip = 0.0
    zp = (C.c_float * 4)()
    assert meao_lib.meao_zbuffer_params(C.byref(p), C.byref(zp)) == L.ERR_INVALID_ARGUMENT


def test_no_cpu_fallback(meao_lib):
    """On a box without a GPU the product path must fail loudly, never compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    cfg = L.Config()
    meao_lib.meao_default_config(C.byref(cfg))
    ctx = C.c_void_p()
    assert meao_lib.meao_create(C.byref(cfg), C.byref(ctx)) == L.ERR_NO_DEVICE
    assert not ctx.value
    from miniengineao_amd import AmbientOcclusion
    with pytest.raises(L.MeaoError) as e:
        AmbientOcclusion(64, 64)
    assert e.value.status == L.ERR_NO_DEVICE


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "miniengineao_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle|#include\s+\"[^\"]*oracle", text, re.M), f


def test_every_variant_arm_names_a_switch_of_the_kernel_source():
    """build.VARIANTS may only carry -D flags whose macro is a switch of the native sources (an arm whose switch has been
    removed builds a copy of the product under another name: four such arms once travelled through an evidence round)."""
    import re
    from miniengineao_amd import build
    csrc = os.path.join(os.path.dirname(build.__file__), "csrc")
    src = "".join(open(os.path.join(csrc, f)).read() for f in sorted(os.listdir(csrc)))
    for name, flags in build.VARIANTS.items():
        for flag in flags:
            macro = re.match(r"-D(\w+)", flag).group(1)
            assert re.search(r"#\s*ifndef\s+%s\b" % macro, src), f"variant {name}: {macro} is not a switch of miniengineao_amd/csrc"


def test_every_kernel_unit_is_part_of_the_unity_file():
    """csrc/meao_kernels.hip (the one-translation-unit form the `clocks` variant and the ISA tools build) includes exactly the
    units the product compiles separately (build.KERNEL_UNITS)."""
    import re
    from miniengineao_amd import build
    csrc = os.path.join(os.path.dirname(build.__file__), "csrc")
    included = re.findall(r'#include "(meao_k_\w+\.hip)"', open(os.path.join(csrc, "meao_kernels.hip")).read())
    assert sorted(included) == sorted(build.KERNEL_UNITS)
    assert sorted(f for f in os.listdir(csrc) if f.startswith("meao_k_") and f.endswith(".hip")) == sorted(build.KERNEL_UNITS)
