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


def component(s, max_batch=1, device=0, **kw):
    """AmbientOcclusion (C ABI) configured exactly like oracle Settings ``s``."""
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
