"""ctypes binding of libmeao_hip.so -- the C ABI declared in include/meao.h.

The library is the product; this module only declares its signatures.  There is no CPU
fallback: if the shared object is missing the import of the hot path fails loudly, and on a
machine without a gfx950 device ``meao_create`` returns MEAO_ERR_NO_DEVICE.

HIP runtime note: a process must hold exactly one HIP runtime.  PyTorch-ROCm wheels bundle their
own ``libamdhip64.so`` (SONAME ``libamdhip64.so.7``, the same as /opt/rocm's).  If torch is
installed, ``load()`` therefore maps torch's copy first (without importing torch), so that our
library's ``NEEDED libamdhip64.so.7`` resolves to it and a later ``import torch`` reuses the same
runtime; loading /opt/rocm's copy first would leave a later-initialised torch without GPUs.
Without torch the RUNPATH of the library (/opt/rocm/lib) is used.
"""
from __future__ import annotations

import ctypes as C
import importlib.util
import os
import sys

_PKG = os.path.dirname(os.path.abspath(__file__))
# MEAO_LIB_PATH: an alternative build of the same library (A/B of kernel variants, tools/run_gpu_variants_ab.sh)
LIB_PATH = os.environ.get("MEAO_LIB_PATH") or os.path.join(_PKG, "lib", "libmeao_hip.so")

ABI_VERSION = 6
MAX_BATCH = 64
NUM_PASSES = 7
PASS_NAMES = ("downsample", "render", "upsample_L4_to_L3", "upsample_L3_to_L2",
              "upsample_L2_to_L1", "upsample_L1_to_L0", "render_hq")
PASS_STREAM_ORDER = (0, 1, 6, 2, 3, 4, 5)     # render_hq runs right after render

OK = 0
ERR_INVALID_ARGUMENT, ERR_HIP, ERR_OUT_OF_MEMORY = -1, -2, -3
ERR_UNSUPPORTED, ERR_NO_DEVICE, ERR_BUFFER_TOO_SMALL = -4, -5, -6
AO_R8, AO_F16 = 0, 1
F16_RTZ_CLAMP, F16_RTNE = 0, 1
MEM_HOST, MEM_DEVICE = 0, 1
DEPTH_F32, DEPTH_UNORM16, DEPTH_UNORM24, DEPTH_F16 = 0, 1, 2, 3
COMPOSITE_MULTIPLY, COMPOSITE_AMBIENT_ONLY, COMPOSITE_DEBUG = 0, 1, 2
FMT_F32, FMT_F16, FMT_UNORM8 = 0, 1, 2
SAMPLES_CHECKER, SAMPLES_EXHAUSTIVE = 0, 1
DEBUG_OCCLUSION_HQ1 = 18
# meao_debug_key (ABI 6: the keys of the launch structures that lost every A/B are gone; fault injection -- meao_test_fail_next_allocs --
# exists only in the `testhooks` variant library, built with -DMEAO_TESTING=1)
(DEBUG_FUSE_COARSE_BLEND, DEBUG_NESTED_MAX_TILES, DEBUG_RENDER_SMALL_MAX_TILES, DEBUG_FINAL_SMALL_MAX_TILES,
 DEBUG_DS_SMALL_MAX_TILES, DEBUG_BLEND_TALL_MIN_TILES, DEBUG_PROFILE_PASS_MASK, DEBUG_NEXT_DOWNSAMPLE_OWN_LAUNCH) = range(8)
POOL_PATH_SAME_DEVICE, POOL_PATH_PEER_DIRECT, POOL_PATH_STAGED = 0, 1, 2
POOL_SPIN_US, POOL_BIND_NUMA = 0, 1
NUM_BUFFERS = 21


class Config(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("device", C.c_int32), ("width", C.c_int32),
                ("height", C.c_int32), ("num_levels", C.c_int32), ("ao_format", C.c_int32),
                ("f16_rounding", C.c_int32), ("max_batch", C.c_int32),
                ("depth_format", C.c_int32), ("hq_levels", C.c_int32), ("sample_set", C.c_int32),
                ("pipelined", C.c_int32)]


class Params(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("noise_filter_tolerance", C.c_float),
                ("blur_tolerance", C.c_float), ("upsample_tolerance", C.c_float),
                ("thickness_modifier", C.c_float), ("intensity", C.c_float),
                ("near_clip", C.c_float), ("far_clip", C.c_float), ("proj00", C.c_float),
                ("reversed_z", C.c_int32), ("single_pass_stereo", C.c_int32)]


class Desc(C.Structure):
    _fields_ = [("debug_id", C.c_int32), ("width", C.c_int32), ("height", C.c_int32),
                ("slices", C.c_int32), ("format", C.c_int32), ("bytes", C.c_uint64)]


class RenderConstants(C.Structure):
    _fields_ = [("inv_thickness_table", C.c_float * 12), ("sample_weight_table", C.c_float * 12),
                ("inv_slice_dimension", C.c_float * 2), ("reject_fadeoff", C.c_float),
                ("intensity", C.c_float)]


class UpsampleConstants(C.Structure):
    _fields_ = [("inv_low_resolution", C.c_float * 2), ("inv_high_resolution", C.c_float * 2),
                ("noise_filter_strength", C.c_float), ("step_size", C.c_float),
                ("blur_tolerance", C.c_float), ("upsample_tolerance", C.c_float)]


# name -> (restype, argtypes); every symbol include/meao.h declares
SIGNATURES = {
    "meao_abi_version": (C.c_int32, []),
    "meao_status_string": (C.c_char_p, [C.c_int32]),
    "meao_default_config": (None, [C.POINTER(Config)]),
    "meao_default_params": (None, [C.POINTER(Params)]),
    "meao_level_dims": (C.c_int32, [C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "meao_zbuffer_params": (C.c_int32, [C.POINTER(Params), C.POINTER(C.c_float * 4)]),
    "meao_render_constants_for": (C.c_int32, [C.c_int32, C.c_int32, C.POINTER(Params), C.c_int32, C.POINTER(RenderConstants)]),
    "meao_render_constants_variant": (C.c_int32, [C.c_int32, C.c_int32, C.POINTER(Params), C.c_int32, C.c_int32,
                                                  C.c_int32, C.POINTER(RenderConstants)]),
    "meao_upsample_constants_for": (C.c_int32, [C.c_int32, C.c_int32, C.POINTER(Params), C.c_int32, C.POINTER(UpsampleConstants)]),
    "meao_describe_buffer": (C.c_int32, [C.POINTER(Config), C.c_int32, C.POINTER(Desc)]),
    "meao_algorithmic_bytes": (C.c_int32, [C.POINTER(Config), C.POINTER(C.c_uint64 * NUM_PASSES)]),
    "meao_create": (C.c_int32, [C.POINTER(Config), C.POINTER(C.c_void_p)]),
    "meao_destroy": (C.c_int32, [C.c_void_p]),
    "meao_resize": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32]),
    "meao_set_params": (C.c_int32, [C.c_void_p, C.POINTER(Params)]),
    "meao_get_params": (C.c_int32, [C.c_void_p, C.POINTER(Params)]),
    "meao_get_config": (C.c_int32, [C.c_void_p, C.POINTER(Config)]),
    "meao_last_error": (C.c_char_p, [C.c_void_p]),
    "meao_execute": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]),
    "meao_execute_batch": (C.c_int32, [C.c_void_p, C.c_int32, C.POINTER(C.c_void_p), C.c_int32,
                                       C.POINTER(C.c_void_p), C.c_int32, C.c_void_p]),
    "meao_prefetch_batch": (C.c_int32, [C.c_void_p, C.c_int32, C.POINTER(C.c_void_p)]),
    "meao_synchronize": (C.c_int32, [C.c_void_p, C.c_void_p]),
    "meao_get_intermediate": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_uint64,
                                          C.c_int32, C.POINTER(Desc)]),
    "meao_set_profiling": (C.c_int32, [C.c_void_p, C.c_int32]),
    "meao_get_pass_times": (C.c_int32, [C.c_void_p, C.POINTER(C.c_float * NUM_PASSES), C.POINTER(C.c_int32)]),
    "meao_selftest": (C.c_int32, [C.c_void_p, C.c_int32, C.POINTER(C.c_uint64)]),
    "meao_set_tracing": (C.c_int32, [C.c_void_p, C.c_int32]),
    "meao_composite_enqueue": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                           C.POINTER(C.c_void_p)]),
    "meao_composite_flush": (C.c_int32, [C.c_void_p, C.c_void_p]),
    "meao_composite_pending": (C.c_int32, [C.c_void_p, C.POINTER(C.c_int32)]),
    "meao_pool_create": (C.c_int32, [C.POINTER(Config), C.POINTER(C.c_int32), C.c_int32, C.POINTER(C.c_void_p)]),
    "meao_pool_destroy": (C.c_int32, [C.c_void_p]),
    "meao_pool_size": (C.c_int32, [C.c_void_p]),
    "meao_pool_context": (C.c_void_p, [C.c_void_p, C.c_int32]),
    "meao_pool_device_of_frame": (C.c_int32, [C.c_void_p, C.c_int32]),
    "meao_pool_last_error": (C.c_char_p, [C.c_void_p]),
    "meao_pool_set_params": (C.c_int32, [C.c_void_p, C.POINTER(Params)]),
    "meao_pool_execute_batch": (C.c_int32, [C.c_void_p, C.c_int32, C.POINTER(C.c_void_p), C.c_int32,
                                            C.POINTER(C.c_void_p), C.c_int32]),
    "meao_pool_gather_to_device": (C.c_int32, [C.c_void_p, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int32]),
    "meao_pool_synchronize": (C.c_int32, [C.c_void_p]),
    "meao_pool_prefetch_batch": (C.c_int32, [C.c_void_p, C.c_int32, C.POINTER(C.c_void_p)]),
    "meao_pool_composite_enqueue": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                                C.POINTER(C.c_void_p)]),
    "meao_pool_composite_flush": (C.c_int32, [C.c_void_p]),
    "meao_pool_composite_pending": (C.c_int32, [C.c_void_p, C.POINTER(C.c_int32)]),
    "meao_pool_gather_path": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32]),
    "meao_device_numa_node": (C.c_int32, [C.c_int32, C.POINTER(C.c_int32), C.c_char_p, C.c_uint64]),
    "meao_pool_configure": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32]),
    "meao_pool_member_placement": (C.c_int32, [C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "meao_hostile_frames": (C.c_int32, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "meao_debug_set": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32]),
    "meao_debug_view": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]),
    "meao_composite": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
}

_lib = None


class MeaoError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"meao status {status}: {message}")
        self.status = status


def _share_torch_hip_runtime() -> None:
    if "torch" in sys.modules:
        return                                  # torch already mapped its runtime
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


def load() -> C.CDLL:
    """Load libmeao_hip.so (built by ``python -m miniengineao_amd.build``).  Raises if absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -m miniengineao_amd.build` "
                "(hipcc, gfx950).  There is no CPU fallback for the SSAO hot path.")
        _share_torch_hip_runtime()
        lib = C.CDLL(LIB_PATH)
        for name, (restype, argtypes) in SIGNATURES.items():
            fn = getattr(lib, name)     # AttributeError if the ABI lost a symbol
            fn.restype = restype
            fn.argtypes = argtypes
        if lib.meao_abi_version() != ABI_VERSION:
            raise ImportError(f"libmeao_hip.so ABI {lib.meao_abi_version()} != binding {ABI_VERSION}")
        _lib = lib
    return _lib


def check(status: int, ctx=None) -> None:
    if status != OK:
        lib = load()
        detail = lib.meao_last_error(ctx).decode() or lib.meao_status_string(status).decode()
        raise MeaoError(status, detail)
