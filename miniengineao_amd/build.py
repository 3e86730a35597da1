"""Builds libmeao_hip.so (the C-ABI product library) in-tree with hipcc for gfx950.

    python -m miniengineao_amd.build [--force] [--verbose]

The library links only against the HIP runtime; there is no torch dependency and no CPU
fallback.  hipcc cross-compiles without a GPU, so this runs in the build container.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

_PKG = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_PKG, "csrc")
_INCLUDE = os.path.join(os.path.dirname(_PKG), "include")
LIB_DIR = os.path.join(_PKG, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libmeao_hip.so")

# The kernels are six translation units over shared device headers (meao_dev*.hpp): they compile in parallel and a
# change to one family rebuilds one unit.  csrc/meao_kernels.hip is the same code as ONE unit (it includes the six):
# variants that need device globals (`clocks`) and the ISA tools build that.
KERNEL_UNITS = ["meao_k_downsample.hip", "meao_k_render.hip", "meao_k_upsample.hip",
                "meao_k_upsample_nested.hip", "meao_k_upsample_fused.hip", "meao_k_misc.hip"]
HOST_UNITS = ["meao_plan.cpp", "meao_api.cpp", "meao_pool.cpp"]
SOURCES = HOST_UNITS + KERNEL_UNITS
HEADERS = ["meao_plan.hpp", "meao_kernels.hpp", "meao_dev.hpp", "meao_dev_downsample.hpp", "meao_dev_render.hpp",
           "meao_dev_upsample.hpp", "meao_dev_blend.hpp", "meao_dev_composite.hpp", "meao_kernels.hip"]
OBJ_DIR = os.path.join(LIB_DIR, "obj")

# -ffp-contract=off: the only fused multiply-adds are the explicit mad()/fma2() calls, which
# is what makes the kernels bit-exact against the oracle.  Correctly rounded '/' and sqrt are
# hipcc's default; the flag is spelled out because parity depends on it.
# -fno-slp-vectorize: the SLP vectorizer pairs scalar f32 ops into v_pk_* at the price of v_mov
# shuffles; on MI355X v_pk_* issues in ~4.7 cycles vs 2 x 3.0, so the shuffles eat the gain
# (A/B on one box: upsample 1-2 % faster without it).  Explicit float2 code still uses v_pk_*.
FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
    "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-slp-vectorize",
    "-fvisibility=hidden", "-Wall", "-Wextra", "-Wno-unused-parameter",
]


COMPILE_ONLY_PREFIXES = ("-O", "-std=", "-ffp-contract", "-fhip-fp32-correctly-rounded", "-fno-slp-vectorize", "-W")


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC)")


def _stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    built = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(_CSRC, f) for f in SOURCES + HEADERS]
    deps += [os.path.join(_INCLUDE, "meao.h"), os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > built for d in deps)


def _run(cmd, verbose):
    if verbose:
        print(" ".join(cmd), flush=True)
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError(f"hipcc failed ({proc.returncode}):\n{' '.join(cmd)}\n{proc.stdout}\n{proc.stderr}")
    if verbose and proc.stderr.strip():
        print(proc.stderr)


def _unit_stale(src: str, obj: str) -> bool:
    """An object is rebuilt when its source, any csrc header, meao.h or this file is newer (headers are few and shared)."""
    if not os.path.exists(obj):
        return True
    built = os.path.getmtime(obj)
    deps = [src, os.path.join(_INCLUDE, "meao.h"), os.path.abspath(__file__)]
    deps += [os.path.join(_CSRC, h) for h in HEADERS if h.endswith(".hpp")]
    if os.path.basename(src) == "meao_kernels.hip":         # the unity unit #includes every kernel unit
        deps += [os.path.join(_CSRC, u) for u in KERNEL_UNITS]
    return any(os.path.getmtime(d) > built for d in deps)


def build_lib(force: bool = False, verbose: bool = False, extra_flags=(), out_path: str = None, unity: bool = False,
              jobs: int = None) -> str:
    """Compiles every unit to an object (in parallel, only the stale ones unless force) and links libmeao_hip.so.
    out_path + extra_flags (-DMEAO_...) build a kernel variant next to the product library (A/B runs); unity=True
    compiles the kernels as the single translation unit csrc/meao_kernels.hip."""
    from concurrent.futures import ThreadPoolExecutor
    if out_path is None and not force and not _stale():
        return LIB_PATH
    final = out_path or LIB_PATH
    tag = "product" if out_path is None else os.path.splitext(os.path.basename(out_path))[0]
    obj_dir = os.path.join(OBJ_DIR, tag)
    os.makedirs(obj_dir, exist_ok=True)
    os.makedirs(os.path.dirname(final), exist_ok=True)
    compile_flags = [f for f in FLAGS if f != "-shared"] + list(extra_flags) + [f"-I{_INCLUDE}"]
    units = HOST_UNITS + (["meao_kernels.hip"] if unity else KERNEL_UNITS)
    flags_stamp = os.path.join(obj_dir, ".flags")
    stamp = " ".join(compile_flags + units)
    same_flags = os.path.exists(flags_stamp) and open(flags_stamp).read() == stamp
    todo, objs = [], []
    for u in units:
        src, obj = os.path.join(_CSRC, u), os.path.join(obj_dir, u + ".o")
        objs.append(obj)
        if force or not same_flags or _unit_stale(src, obj):
            todo.append([hipcc(), *compile_flags, "-c", src, "-o", obj])
    jobs = jobs or max(1, min(len(todo), os.cpu_count() or 1))
    if todo:
        with ThreadPoolExecutor(jobs) as ex:
            list(ex.map(lambda c: _run(c, verbose), todo))
    open(flags_stamp, "w").write(stamp)
    tmp = final + ".tmp"
    # the link line comes from FLAGS too: everything but what only means something to the compiler proper
    link_flags = [f for f in FLAGS if not f.startswith(COMPILE_ONLY_PREFIXES)]
    _run([hipcc(), *link_flags, *objs, "-o", tmp], verbose)
    os.replace(tmp, final)
    return final


# Experimental arms of meao_kernels.hip (MEAO_X_* switches) that are kept in the source: name -> -D flags
UNITY_VARIANTS = ("clocks",)                        # device globals shared by all kernels: one translation unit
VARIANTS = {
    "clocks": ["-DMEAO_X_PHASE_CLOCKS=1"],          # diagnostic: phase stamps, render residency log
    "exactr8": ["-DMEAO_X_UPS_EXACT_R8=1"],         # every UNORM8 bilateral result through the exact-division sequences (cross-check of the estimate)
    "testhooks": ["-DMEAO_TESTING=1"],              # the product + meao_test_fail_next_allocs (fault injection for the resize tests; not in the product ABI)
    "pair": ["-DMEAO_X_BIL_PAIR_RCP=1"],            # three reciprocals per UNORM8 bilateral texel instead of five (round-5 A/B: no gain in the kernels)
    "ntstore": ["-DMEAO_X_FINAL_NT_STORE=1"],       # non-temporal stores for the result texels of the full-resolution pass (the form of rounds 2-4)
    "lowbuf": ["-DMEAO_X_LOWDEPTH_FROM_RAW=0"],     # the full-resolution pass reads its LoResDB window from the LowDepth1 buffer (the form up to round 6a)
    "nowt": ["-DMEAO_X_BIL_WHOLE_TILE=0"],          # without the unmasked copy of the bilateral phase (the round-3 form of the upsample tile)
}


def build_variants(names=None, strict=False):
    """Variant libraries miniengineao_amd/lib/variants/libmeao_<name>.so (MEAO_LIB_PATH selects one at run time).
    Experimental, off-by-default arms: a variant that fails to compile is reported and skipped (its stale .so is
    removed, so tests/test_variants_gpu.py does not run an old build) -- only the product library gates a build
    (ADVICE r3).  strict=True raises instead.  Returns the paths that were built."""
    from concurrent.futures import ThreadPoolExecutor
    out_dir = os.path.join(LIB_DIR, "variants")
    os.makedirs(out_dir, exist_ok=True)
    todo = [(n, f) for n, f in VARIANTS.items() if names is None or n in names]
    if names is None:           # libraries of arms that no longer exist
        for old in os.listdir(out_dir):
            if old.startswith("libmeao_") and old[len("libmeao_"):-len(".so")] not in VARIANTS:
                os.remove(os.path.join(out_dir, old))

    def one(nf):
        path = os.path.join(out_dir, f"libmeao_{nf[0]}.so")
        try:
            return build_lib(force=True, extra_flags=nf[1], out_path=path, unity=nf[0] in UNITY_VARIANTS, jobs=2)
        except RuntimeError as e:
            if strict:
                raise
            if os.path.exists(path):
                os.remove(path)
            print(f"[build] variant {nf[0]} {nf[1]} did NOT build (skipped, the product is unaffected):\n{str(e)[-2000:]}",
                  file=sys.stderr, flush=True)
            return None

    with ThreadPoolExecutor(4) as ex:
        return [p for p in ex.map(one, todo) if p]


DEMO_PATH = os.path.join(LIB_DIR, "ao_host_demo")


def build_host_demo(force: bool = False) -> str:
    """examples/ao_host_demo.cpp: plain g++ host program over include/meao.hpp + the .so."""
    src = os.path.join(os.path.dirname(_PKG), "examples", "ao_host_demo.cpp")
    build_lib()
    deps = [src, os.path.join(_INCLUDE, "meao.hpp"), os.path.join(_INCLUDE, "meao.h")]
    if not force and os.path.exists(DEMO_PATH) and all(os.path.getmtime(d) <= os.path.getmtime(DEMO_PATH) for d in deps):
        return DEMO_PATH
    cmd = ["g++", "-O2", "-std=c++17", "-Wall", "-Wextra", f"-I{_INCLUDE}", src, "-o", DEMO_PATH,
           f"-L{LIB_DIR}", "-lmeao_hip", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath-link," + "/opt/rocm/lib"]
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError(f"g++ failed:\n{proc.stdout}\n{proc.stderr}")
    return DEMO_PATH


def build_tools(force: bool = False):
    """tools/*.hip: the microbenchmarks whose results DESIGN.md quotes (VALU issue rates, exactness of
    the v_rcp_f32 division sequences).  Stand-alone HIP programs, run on the GPU box."""
    out = []
    tools = os.path.join(os.path.dirname(_PKG), "tools")
    for name, extra in (("ubench_valu", []), ("ubench_issue", ["-Wno-unused-value"]), ("ubench_lds", []), ("ubench_launch", []), ("ubench_fetch", []),
                        ("ubench_div", ["-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt"])):
        src, dst = os.path.join(tools, name + ".hip"), os.path.join(LIB_DIR, name)
        if not force and os.path.exists(dst) and os.path.getmtime(dst) >= os.path.getmtime(src):
            out.append(dst)
            continue
        os.makedirs(LIB_DIR, exist_ok=True)
        proc = subprocess.run([hipcc(), "--offload-arch=gfx950", "-O2", *extra, src, "-o", dst], capture_output=True, text=True)
        if proc.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{proc.stderr}")
        out.append(dst)
    return out


if __name__ == "__main__":
    path = build_lib(force="--force" in sys.argv, verbose="--verbose" in sys.argv or "-v" in sys.argv)
    print(path)
    print(build_host_demo(force="--force" in sys.argv))
    print(*build_tools(force="--force" in sys.argv))
