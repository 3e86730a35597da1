"""ctypes front-end of the CPU oracle (oracle/libmeao_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package (miniengineao_amd) never
imports this module.  Parity against reference *outputs* is unpinned (the
reference has no goldens and cannot run here); see meao_oracle.h.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libmeao_oracle.so")

AO_R8, AO_F16 = 0, 1
DEPTH_F32, DEPTH_UNORM16, DEPTH_UNORM24, DEPTH_F16 = 0, 1, 2, 3
DEPTH_DTYPE = {0: np.float32, 1: np.uint16, 2: np.uint32, 3: np.uint16}
F16_RTZ, F16_RTNE = 0, 1

# debug ids of AmbientOcclusion.cs:789-808
BUFFER_IDS = {
    1: "linear_depth", 2: "low_depth1", 3: "low_depth2", 4: "low_depth3", 5: "low_depth4",
    6: "tiled_depth1", 7: "tiled_depth2", 8: "tiled_depth3", 9: "tiled_depth4",
    10: "occlusion1", 11: "occlusion2", 12: "occlusion3", 13: "occlusion4",
    14: "combined1", 15: "combined2", 16: "combined3", 17: "result",
}
# not in the reference's _debug list: the Render.main (wide) targets of the hq_levels variant
HQ_BUFFER_IDS = {18: "occlusion_hq1", 19: "occlusion_hq2", 20: "occlusion_hq3", 21: "occlusion_hq4"}
SAMPLES_CHECKER, SAMPLES_EXHAUSTIVE = 0, 1


class Desc(C.Structure):
    _fields_ = [
        ("width", C.c_int32), ("height", C.c_int32), ("num_levels", C.c_int32),
        ("ao_format", C.c_int32), ("f16_rounding", C.c_int32), ("reversed_z", C.c_int32),
        ("noise_filter_tolerance", C.c_float), ("blur_tolerance", C.c_float),
        ("upsample_tolerance", C.c_float), ("thickness_modifier", C.c_float),
        ("intensity", C.c_float), ("near_clip", C.c_float), ("far_clip", C.c_float),
        ("proj00", C.c_float), ("depth_format", C.c_int32),
        ("single_pass_stereo", C.c_int32), ("hq_levels", C.c_int32), ("sample_set", C.c_int32),
    ]


class Buffers(C.Structure):
    _fields_ = [
        ("linear_depth", C.c_void_p), ("low_depth", C.c_void_p * 4),
        ("tiled_depth", C.c_void_p * 4), ("occlusion", C.c_void_p * 4),
        ("combined", C.c_void_p * 3), ("result", C.c_void_p), ("occlusion_hq", C.c_void_p * 4),
    ]


class RenderConsts(C.Structure):
    _fields_ = [("inv_thickness", C.c_float * 12), ("sample_weight", C.c_float * 12),
                ("inv_slice_dim", C.c_float * 2), ("reject_fadeoff", C.c_float),
                ("intensity", C.c_float)]


class UpsampleConsts(C.Structure):
    _fields_ = [("inv_low_res", C.c_float * 2), ("inv_high_res", C.c_float * 2),
                ("noise_filter_strength", C.c_float), ("step_size", C.c_float),
                ("blur_tolerance", C.c_float), ("upsample_tolerance", C.c_float)]


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (seconds).  Building the checker is not using it."""
    srcs = [os.path.join(_HERE, f) for f in ("meao_oracle.c", "meao_hlsl_emul.c", "meao_oracle.h", "Makefile")]
    stale = force or not os.path.exists(_LIB_PATH) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs)
    if stale:
        subprocess.run(["make", "-C", _HERE, "-B", "libmeao_oracle.so"], check=True,
                       stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        L.meao_oracle_run.argtypes = [C.POINTER(Desc), C.c_void_p, C.POINTER(Buffers), C.c_int32]
        L.meao_oracle_run.restype = C.c_int32
        L.meao_hlsl_emul_run.argtypes = [C.POINTER(Desc), C.c_void_p, C.POINTER(Buffers)]
        L.meao_hlsl_emul_run.restype = C.c_int32
        L.meao_oracle_level_dims.argtypes = [C.c_int32, C.c_int32, C.c_int32,
                                             C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.meao_oracle_render_constants.argtypes = [C.POINTER(Desc), C.c_int32, C.POINTER(RenderConsts)]
        L.meao_oracle_render_constants_hq.argtypes = [C.POINTER(Desc), C.c_int32, C.POINTER(RenderConsts)]
        L.meao_oracle_upsample_constants.argtypes = [C.POINTER(Desc), C.c_int32, C.POINTER(UpsampleConsts)]
        L.meao_oracle_zbuffer_params.argtypes = [C.POINTER(Desc), C.POINTER(C.c_float * 4)]
        L.meao_oracle_sample_thickness.argtypes = [C.POINTER(C.c_float * 12)]
        L.meao_oracle_f32_to_f16.argtypes = [C.c_float, C.c_int32]
        L.meao_oracle_f32_to_f16.restype = C.c_uint16
        L.meao_oracle_f16_to_f32.argtypes = [C.c_uint16]
        L.meao_oracle_f16_to_f32.restype = C.c_float
        L.meao_oracle_f32_to_unorm8.argtypes = [C.c_float]
        L.meao_oracle_f32_to_unorm8.restype = C.c_uint8
        L.meao_oracle_unorm8_to_f32.argtypes = [C.c_uint8]
        L.meao_oracle_unorm8_to_f32.restype = C.c_float
        _lib = L
    return _lib


@dataclass
class Settings:
    """Component properties (AmbientOcclusion.cs:20-68) + the camera terms the path reads."""
    width: int
    height: int
    num_levels: int = 4
    ao_format: int = AO_R8
    f16_rounding: int = F16_RTZ
    reversed_z: bool = True
    noise_filter_tolerance: float = 0.0
    blur_tolerance: float = -4.6
    upsample_tolerance: float = -12.0
    thickness_modifier: float = 1.0
    intensity: float = 1.0
    near_clip: float = 0.1
    far_clip: float = 100.0
    proj00: float = 1.0
    depth_format: int = DEPTH_F32
    single_pass_stereo: bool = False     # AO.cs:392-401,680
    hq_levels: int = 0                   # coarsest N levels also run Render.main + main_premin*
    sample_set: int = SAMPLES_CHECKER    # or SAMPLES_EXHAUSTIVE (REN:144-159)

    def hq_level_list(self):
        return [k for k in range(1, self.num_levels + 1) if k > self.num_levels - self.hq_levels]

    def desc(self) -> Desc:
        return Desc(self.width, self.height, self.num_levels, self.ao_format, self.f16_rounding,
                    1 if self.reversed_z else 0, self.noise_filter_tolerance, self.blur_tolerance,
                    self.upsample_tolerance, self.thickness_modifier, self.intensity,
                    self.near_clip, self.far_clip, self.proj00, self.depth_format,
                    1 if self.single_pass_stereo else 0, self.hq_levels, self.sample_set)


def level_dims(width: int, height: int, level: int):
    div = 1 << level
    return (width + div - 1) // div, (height + div - 1) // div


def allocate(s: Settings):
    """The 17 buffers of AO.cs:453-475 as numpy arrays, keyed like BUFFER_IDS."""
    ao_dt = np.uint8 if s.ao_format == AO_R8 else np.uint16
    dims = [level_dims(s.width, s.height, k) for k in range(7)]
    out = {"linear_depth": np.zeros((dims[0][1], dims[0][0]), np.uint16)}
    for k in range(1, 5):
        out[f"low_depth{k}"] = np.zeros((dims[k][1], dims[k][0]), np.float32)
        out[f"tiled_depth{k}"] = np.zeros((16, dims[k + 2][1], dims[k + 2][0]), np.uint16)
        out[f"occlusion{k}"] = np.zeros((dims[k][1], dims[k][0]), ao_dt)
        if k <= 3:
            out[f"combined{k}"] = np.zeros((dims[k][1], dims[k][0]), ao_dt)
    out["result"] = np.zeros((dims[0][1], dims[0][0]), ao_dt)
    for k in s.hq_level_list():
        out[f"occlusion_hq{k}"] = np.zeros((dims[k][1], dims[k][0]), ao_dt)
    return out


def _buffers(arrs) -> Buffers:
    b = Buffers()
    b.linear_depth = arrs["linear_depth"].ctypes.data
    for k in range(4):
        b.low_depth[k] = arrs[f"low_depth{k + 1}"].ctypes.data
        b.tiled_depth[k] = arrs[f"tiled_depth{k + 1}"].ctypes.data
        b.occlusion[k] = arrs[f"occlusion{k + 1}"].ctypes.data
        if k < 3:
            b.combined[k] = arrs[f"combined{k + 1}"].ctypes.data
    b.result = arrs["result"].ctypes.data
    for k in range(4):
        if f"occlusion_hq{k + 1}" in arrs:
            b.occlusion_hq[k] = arrs[f"occlusion_hq{k + 1}"].ctypes.data
    return b


def host_cores() -> int:
    """CPUs this process may actually use: the affinity mask, capped by the cgroup CPU quota (a container on a 256-thread host
    may own 16 of them -- `os.cpu_count()` still says 256, and 256 busy threads on a 16-CPU quota run slower than 16)."""
    import math
    import os
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    try:                                                     # cgroup v2
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, math.ceil(int(quota) / int(period))))
    except (OSError, ValueError):
        try:                                                 # cgroup v1
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0 and period > 0:
                n = min(n, max(1, math.ceil(quota / period)))
        except (OSError, ValueError):
            pass
    return max(1, n)


def run(depth: np.ndarray, s: Settings, nthreads: int = 1, emulate_hlsl: bool = False,
        result_only: bool = False):
    """Run the oracle; returns dict name -> array (all 17 buffers, or just 'result')."""
    depth = np.ascontiguousarray(depth, dtype=DEPTH_DTYPE[s.depth_format])
    assert depth.shape == (s.height, s.width), (depth.shape, s.height, s.width)
    d = s.desc()
    if result_only:
        ao_dt = np.uint8 if s.ao_format == AO_R8 else np.uint16
        arrs = {"result": np.zeros((s.height, s.width), ao_dt)}
        b = Buffers()
        b.result = arrs["result"].ctypes.data
    else:
        arrs = allocate(s)
        b = _buffers(arrs)
    if emulate_hlsl:
        rc = lib().meao_hlsl_emul_run(C.byref(d), depth.ctypes.data, C.byref(b))
    else:
        rc = lib().meao_oracle_run(C.byref(d), depth.ctypes.data, C.byref(b), int(nthreads))
    if rc != 0:
        raise RuntimeError(f"oracle failed: {rc}")
    return arrs


def render_constants(s: Settings, level: int) -> RenderConsts:
    out = RenderConsts()
    d = s.desc()
    lib().meao_oracle_render_constants(C.byref(d), level, C.byref(out))
    return out


def render_constants_hq(s: Settings, level: int) -> RenderConsts:
    out = RenderConsts()
    d = s.desc()
    lib().meao_oracle_render_constants_hq(C.byref(d), level, C.byref(out))
    return out


def upsample_constants(s: Settings, low_level: int) -> UpsampleConsts:
    out = UpsampleConsts()
    d = s.desc()
    lib().meao_oracle_upsample_constants(C.byref(d), low_level, C.byref(out))
    return out


def zbuffer_params(s: Settings):
    out = (C.c_float * 4)()
    d = s.desc()
    lib().meao_oracle_zbuffer_params(C.byref(d), C.byref(out))
    return list(out)


def sample_thickness():
    out = (C.c_float * 12)()
    lib().meao_oracle_sample_thickness(C.byref(out))
    return np.array(list(out), dtype=np.float32)


def encode_depth(raw: np.ndarray, depth_format: int) -> np.ndarray:
    """Quantise float raw depth in [0,1] into a depth-buffer storage format (test inputs)."""
    raw = np.asarray(raw, dtype=np.float64)
    if depth_format == DEPTH_UNORM16:
        return np.rint(raw * 65535.0).astype(np.uint16)
    if depth_format == DEPTH_UNORM24:     # D24S8: stencil garbage in the high byte must be ignored
        code = np.rint(raw * 16777215.0).astype(np.uint32)
        return code | (np.uint32(0xA5) << np.uint32(24))
    if depth_format == DEPTH_F16:
        return raw.astype(np.float16).view(np.uint16)
    return raw.astype(np.float32)


def composite(ao: np.ndarray, color_rgba16f: np.ndarray, mode: int, ao_format: int = AO_R8,
              gbuffer0_rgba8: np.ndarray = None):
    """In place; mode 0 multiply, 1 ambient-only, 2 debug (Blit.shader passes 2, 1, 3)."""
    h, w = ao.shape
    L = lib()
    L.meao_oracle_composite.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    L.meao_oracle_composite.restype = C.c_int32
    rc = L.meao_oracle_composite(w, h, ao_format, mode, ao.ctypes.data, color_rgba16f.ctypes.data,
                                 gbuffer0_rgba8.ctypes.data if gbuffer0_rgba8 is not None else None)
    if rc != 0:
        raise RuntimeError(f"oracle composite failed: {rc}")


def debug_view(buffers: dict, debug_id: int, s: "Settings") -> np.ndarray:
    """PushDebugBlitCommands (AO.cs:787-820) over the oracle's buffers: cmd.Blit(rt, _result) point
    sampling for 2D buffers, the 4x4 slice grid of Blit.shader:136-155 for tiled arrays; sampling
    positions in exact integer arithmetic; the store converts like every AO store."""
    src = buffers[BUFFER_IDS[debug_id]]
    W, H = s.width, s.height
    x, y = np.arange(W, dtype=np.int64)[None, :], np.arange(H, dtype=np.int64)[:, None]
    if src.ndim == 2:
        sh, sw = src.shape
        vals = src[((2 * y + 1) * sh) // (2 * H), ((2 * x + 1) * sw) // (2 * W)]
    else:
        _, sh, sw = src.shape
        nx, ny = 4 * x + 2, 4 * y + 2
        vals = src[nx // W + 4 * (ny // H), ((ny % H) * sh) // H, ((nx % W) * sw) // W]
    if src.dtype == np.float32:
        f = vals
    elif src.dtype == np.uint8:
        f = vals.astype(np.float32) / np.float32(255)
    else:
        f = f16_bits_to_f32(vals)
    L = lib()
    if s.ao_format == AO_R8:
        enc = np.vectorize(lambda v: L.meao_oracle_f32_to_unorm8(float(v)), otypes=[np.uint8])
    else:
        enc = np.vectorize(lambda v: L.meao_oracle_f32_to_f16(float(v), s.f16_rounding), otypes=[np.uint16])
    return enc(f)


def f16_bits_to_f32(bits: np.ndarray) -> np.ndarray:
    return np.asarray(bits, dtype=np.uint16).view(np.float16).astype(np.float32)
