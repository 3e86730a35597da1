"""Shared helpers of the parity tests: one Settings object drives both the oracle and the
C-ABI library so that every comparison runs on identical inputs and parameters."""
from __future__ import annotations

import numpy as np

from miniengineao_amd import synth

NAMES = {1: "linear_depth", 2: "low_depth1", 3: "low_depth2", 4: "low_depth3", 5: "low_depth4",
         6: "tiled_depth1", 7: "tiled_depth2", 8: "tiled_depth3", 9: "tiled_depth4",
         10: "occlusion1", 11: "occlusion2", 12: "occlusion3", 13: "occlusion4",
         14: "combined1", 15: "combined2", 16: "combined3", 17: "result",
         18: "occlusion_hq1", 19: "occlusion_hq2", 20: "occlusion_hq3", 21: "occlusion_hq4"}


def settings(O, w, h, cam=synth.DEFAULT_CAMERA, **kw):
    return O.Settings(w, h, proj00=cam.proj00(w, h), near_clip=cam.near, far_clip=cam.far,
                      reversed_z=cam.reversed_z, **kw)


def component(s, max_batch=1, device=0, debug=None, **kw):
    """AmbientOcclusion (C ABI) configured exactly like oracle Settings ``s``.  ``debug``: {meao_debug_key: value}
    launch-structure overrides applied through meao_debug_set."""
    from miniengineao_amd import AmbientOcclusion
    ao = AmbientOcclusion(s.width, s.height, device=device, num_levels=s.num_levels,
                          ao_format=s.ao_format, f16_rounding=s.f16_rounding, max_batch=max_batch,
                          near_clip=s.near_clip, far_clip=s.far_clip, projection00=s.proj00,
                          reversed_z=s.reversed_z, hq_levels=s.hq_levels, sample_set=s.sample_set,
                          single_pass_stereo=s.single_pass_stereo, **kw)
    ao.noiseFilterTolerance = s.noise_filter_tolerance
    ao.blurTolerance = s.blur_tolerance
    ao.upsampleTolerance = s.upsample_tolerance
    ao.thicknessModifier = s.thickness_modifier
    ao.intensity = s.intensity
    for key, value in (debug or {}).items():
        ao.debug_set(key, value)
    return ao


def valid_debug_ids(num_levels, hq_levels=0):
    ids = list(range(1, 10))
    ids += [10 + k for k in range(num_levels)]
    ids += [14 + k for k in range(num_levels - 1)]
    ids.append(17)
    ids += [17 + k for k in range(1, num_levels + 1) if k > num_levels - hq_levels]
    return ids


def diff_report(name, got, want):
    bad = np.argwhere(got != want)
    first = tuple(bad[0])
    return (f"{name}: {len(bad)} of {got.size} texels differ; first at {first}: "
            f"got {got[first]!r} want {want[first]!r}")


def checksum(arr: np.ndarray) -> int:
    """Order-sensitive 64-bit checksum (FNV-style fold of 8-byte words) used for goldens."""
    b = np.ascontiguousarray(arr).view(np.uint8).ravel()
    pad = (-len(b)) % 8
    if pad:
        b = np.concatenate([b, np.zeros(pad, np.uint8)])
    w = b.view(np.uint64)
    idx = np.arange(1, len(w) + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        mixed = (w ^ (idx * np.uint64(0x9E3779B97F4A7C15))) * np.uint64(0x100000001B3)
        return int(np.bitwise_xor.reduce(mixed) ^ np.uint64(len(b)))


# ---- hostile depth inputs (tests/test_hostile_depth.py, tools/fuzz_gpu.py)

def nan_aware_equal(got, want):
    if got.dtype == np.float32:
        g, w = got.view(np.uint32), want.view(np.uint32)
        gn = (g & 0x7fffffff) > 0x7f800000
        wn = (w & 0x7fffffff) > 0x7f800000
    elif got.dtype == np.uint16:      # f16 bit patterns (depth mips, fp16 AO)
        g, w = got, want
        gn = (g & 0x7fff) > 0x7c00
        wn = (w & 0x7fff) > 0x7c00
    else:
        return np.array_equal(got, want), got != want
    bad = ~((g == w) | (gn & wn))
    return not bad.any(), bad


def hostile_frame(w, h, seed, cam=synth.DEFAULT_CAMERA, density=0.01, kinds=None):
    """S2 frame with hostile texels sprinkled in (isolated ones and small blocks)."""
    rng = np.random.default_rng(seed)
    d = synth.make("S2", w, h, seed=seed).copy()
    fpn = np.float32(cam.far) / np.float32(cam.near)
    zp0 = (fpn - np.float32(1)) if cam.reversed_z else (np.float32(1) - fpn)
    zp1 = np.float32(1) if cam.reversed_z else fpn
    zero_den = np.float32(-zp1 / zp0)          # ZBufferParams.x * d + ZBufferParams.y == 0 (or nearly)
    values = {
        "nan": np.float32(np.nan), "pinf": np.float32(np.inf), "ninf": np.float32(-np.inf),
        "neg": np.float32(-0.25), "big": np.float32(7.5), "huge": np.float32(3e38), "nhuge": np.float32(-3e38),
        "denorm": np.float32(1e-41), "negzero": np.float32(-0.0), "zero_den": zero_den,
        "tiny_den": np.nextafter(zero_den, np.float32(0), dtype=np.float32), "one": np.float32(1.0),
        "zero": np.float32(0.0),
    }
    names = list(values) if kinds is None else list(kinds)
    n = max(1, int(w * h * density))
    ys, xs = rng.integers(0, h, n), rng.integers(0, w, n)
    for i in range(n):
        v = values[names[i % len(names)]]
        if i % 7 == 0:
            d[ys[i]:ys[i] + 3, xs[i]:xs[i] + 5] = v
        else:
            d[ys[i], xs[i]] = v
    return d




def run_against_testhooks(script):
    """Run tests/<script> in its own process against the `testhooks` variant library (-DMEAO_TESTING=1: the only build that exports
    meao_test_*).  The library is built on the spot when it is missing -- a GPU run must never skip these checks silently
     -- and its absence after that is a failure, not a skip."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "miniengineao_amd", "lib", "variants", "libmeao_testhooks.so")
    if not os.path.exists(lib):
        from miniengineao_amd import build
        build.build_variants(["testhooks"], strict=True)
    assert os.path.exists(lib), "the testhooks variant library could not be built"
    proc = subprocess.run([sys.executable, os.path.join(root, "tests", script)], cwd=root,
                          env=dict(os.environ, MEAO_LIB_PATH=lib), capture_output=True, text=True, timeout=900)
    assert proc.returncode == 0, (proc.stdout[-1500:], proc.stderr[-1500:])
