"""Host-side mirrors of the reference component: the Python class, the C++ header-only class
and the (uncompilable here) C# P/Invoke source all expose AmbientOcclusion.cs's surface and
bind exactly the C ABI."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
with open(os.path.join(ROOT, "include", "meao.h")) as _f:
    HEADER = _f.read()
REFERENCE_PROPERTIES = ["noiseFilterTolerance", "blurTolerance", "upsampleTolerance",
                        "thicknessModifier", "intensity", "ambientOnly"]     # AmbientOcclusion.cs:22-66


def c_prototypes():
    protos = {}
    for m in re.finditer(r"MEAO_API\s+[\w\s\*]+?\b(meao_\w+)\s*\((.*?)\);", HEADER, re.S):
        args = m.group(2).strip()
        protos[m.group(1)] = 0 if args == "void" else len(args.split(","))
    return protos


def test_python_mirror_has_the_reference_properties():
    from miniengineao_amd.ambient_occlusion import AmbientOcclusion
    for name in REFERENCE_PROPERTIES:
        assert isinstance(getattr(AmbientOcclusion, name), property), name


def test_csharp_dllimports_match_the_header():
    src = open(os.path.join(ROOT, "bindings", "csharp", "MeaoNative.cs")).read()
    imports = {}
    for m in re.finditer(r"\[DllImport\(Lib\)\]\s*public static extern \w+ (meao_\w+)\((.*?)\);", src):
        args = m.group(2).strip()
        imports[m.group(1)] = 0 if not args else len(args.split(","))
    assert imports == c_prototypes()
    assert 'const string Lib = "meao_hip"' in src
    # struct field order mirrors the C structs
    for cname, csname in (("meao_config", "MeaoConfig"), ("meao_params", "MeaoParams"), ("meao_desc", "MeaoDesc")):
        cbody = re.sub(r"/\*.*?\*/", "", re.search(r"typedef struct %s \{(.*?)\}" % cname, HEADER, re.S).group(1), flags=re.S)
        cnames = []
        for decl in cbody.split(";"):
            if decl.strip():
                cnames += [n.strip() for n in decl.strip().split(None, 1)[1].split(",")]
        csbody = re.search(r"public struct %s\s*\{(.*?)\}" % csname, src, re.S).group(1)
        assert re.findall(r"public \w+ (\w+);", csbody) == cnames, cname


def test_csharp_wrapper_keeps_the_reference_surface():
    src = open(os.path.join(ROOT, "bindings", "csharp", "AmbientOcclusionOverMeao.cs")).read()
    assert "namespace MiniEngineAO" in src and "public sealed class AmbientOcclusion" in src
    for name in REFERENCE_PROPERTIES:
        assert re.search(r"public (float|bool) %s\b" % name, src), name
    # same defaults as the reference's serialized fields
    for field, default in (("_noiseFilterTolerance", "0"), ("_blurTolerance", "-4.6f"), ("_upsampleTolerance", "-12"),
                           ("_thicknessModifier", "1"), ("_intensity", "1"), ("_ambientOnly", "true")):
        assert re.search(r"%s = %s;" % (field, re.escape(default)), src), field


def test_cpp_header_compiles_and_keeps_the_surface(tmp_path):
    src = tmp_path / "t.cpp"
    src.write_text('#include "meao.hpp"\n'
                   "void probe(MiniEngineAO::AmbientOcclusion &ao) {\n"
                   "  ao.noiseFilterTolerance = ao.blurTolerance = ao.upsampleTolerance = -2.0f;\n"
                   "  ao.thicknessModifier = 2.0f; ao.intensity = 1.5f; ao.ambientOnly = false;\n"
                   "  ao.Render(nullptr, nullptr); ao.Resize(8, 8); (void)ao.HostileFrames(); }\n"
                   "void probe_pool(MiniEngineAO::AmbientOcclusionPool &pool) {\n"
                   "  pool.PrefetchBatch({}); pool.RenderDeviceBatch({}, {}); pool.GatherToDevice({}, {}, 0);\n"
                   "  (void)pool.GatherPath(0, 0); pool.Synchronize(); (void)pool.Size(); }\n")
    subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Wextra", "-Werror",
                    "-I" + os.path.join(ROOT, "include"), str(src)], check=True)
    # the C header must also be valid C
    csrc = tmp_path / "t.c"
    csrc.write_text('#include "meao.h"\nint main(void) { meao_config c; meao_default_config(&c); return (int)c.struct_size; }\n')
    subprocess.run(["gcc", "-std=c11", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-pedantic",
                    "-I" + os.path.join(ROOT, "include"), str(csrc)], check=True)
