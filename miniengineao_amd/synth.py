"""Synthetic raw-depth inputs for the SSAO hot path.

All generators return a float32 (H, W) array of *raw device depth* in (0, 1)
strictly (no sky texels), reversed-Z convention: Linear01 depth L maps to
d = (1/L - 1) / (far/near - 1), so that Linearize (Downsample1.compute:37-48)
with ZBufferParams = (far/near - 1, 1) gives back L.  Only + - * / sqrt in
float64 followed by one rounding to float32 are used, so any language
reproduces the bits.

  S1  radial gradient                       (BASELINE config 1, base of 3/5)
  S2  S1 + seeded rectangles/discs + dither (configs 3, 4, 5)
  S3  analytic "Sponza-like" atrium         (config 2; sponza.obj is a missing blob)
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np


@dataclass(frozen=True)
class Camera:
    """Camera terms the path reads (AmbientOcclusion.cs:563-573)."""
    near: float = 0.1
    far: float = 100.0
    fov_y_deg: float = 60.0
    reversed_z: bool = True

    def proj00(self, width: int, height: int) -> float:
        # Unity Matrix4x4.Perspective: m00 = cot(fovY/2) / aspect
        aspect = width / height
        return float(np.float32(1.0 / (math.tan(math.radians(self.fov_y_deg) * 0.5) * aspect)))


DEFAULT_CAMERA = Camera()
SPONZA_CAMERA = Camera(near=0.01, far=100.0, fov_y_deg=30.0)   # Sponza.unity:921-923


def linear01_to_raw(lin: np.ndarray, cam: Camera = DEFAULT_CAMERA) -> np.ndarray:
    """Invert Linearize for reversed / conventional Z; float64 in, float32 out."""
    fpn = float(np.float32(cam.far) / np.float32(cam.near))
    lin = np.asarray(lin, dtype=np.float64)
    if cam.reversed_z:
        raw = (1.0 / lin - 1.0) / (fpn - 1.0)
    else:
        raw = (1.0 / lin - fpn) / (1.0 - fpn)
    raw32 = raw.astype(np.float32)
    # keep 0 < d < 1 strictly: no sky texels in graded inputs
    tiny = np.float32(1e-7)
    return np.clip(raw32, tiny, np.float32(1.0) - np.float32(6e-8)).astype(np.float32)


def _radial_linear01(width: int, height: int) -> np.ndarray:
    cx, cy = (width - 1) / 2.0, (height - 1) / 2.0
    x = np.arange(width, dtype=np.float64)[None, :]
    y = np.arange(height, dtype=np.float64)[:, None]
    norm = math.sqrt(cx * cx + cy * cy) if (cx or cy) else 1.0
    r = np.sqrt((x - cx) ** 2 + (y - cy) ** 2) / norm
    return 0.05 + 0.90 * r


def radial_gradient(width: int, height: int, cam: Camera = DEFAULT_CAMERA) -> np.ndarray:
    """S1: Linear01 = 0.05 + 0.90 r, r = normalised distance from the image centre."""
    return linear01_to_raw(_radial_linear01(width, height), cam)


class _XorShift32:
    def __init__(self, seed: int):
        self.s = (seed & 0xFFFFFFFF) or 0x1234ABCD

    def next(self) -> int:
        s = self.s
        s ^= (s << 13) & 0xFFFFFFFF
        s ^= s >> 17
        s ^= (s << 5) & 0xFFFFFFFF
        self.s = s
        return s

    def unit(self) -> float:
        return self.next() / 4294967296.0


def occluder_field(width: int, height: int, seed: int = 0x1234ABCD, n_rects: int = 256,
                   n_discs: int = 256, cam: Camera = DEFAULT_CAMERA) -> np.ndarray:
    """S2: S1 plus seeded axis-aligned rectangles and discs (nearer than what they
    cover) plus a +-1e-4 relative per-pixel hash dither: edges, plateaus, rejections."""
    lin = _radial_linear01(width, height)
    rng = _XorShift32(seed)
    scale = min(width, height)
    for _ in range(n_rects):
        cx, cy = rng.unit() * width, rng.unit() * height
        hw = (0.01 + 0.09 * rng.unit()) * scale
        hh = (0.01 + 0.09 * rng.unit()) * scale
        frac = 0.35 + 0.6 * rng.unit()
        x0, x1 = max(0, int(cx - hw)), min(width, int(cx + hw) + 1)
        y0, y1 = max(0, int(cy - hh)), min(height, int(cy + hh) + 1)
        if x0 >= x1 or y0 >= y1:
            continue
        region = lin[y0:y1, x0:x1]
        lin[y0:y1, x0:x1] = np.minimum(region, region.min() * frac + 0.02 * (1 - frac))
    for _ in range(n_discs):
        cx, cy = rng.unit() * width, rng.unit() * height
        rad = (0.01 + 0.07 * rng.unit()) * scale
        frac = 0.35 + 0.6 * rng.unit()
        x0, x1 = max(0, int(cx - rad)), min(width, int(cx + rad) + 1)
        y0, y1 = max(0, int(cy - rad)), min(height, int(cy + rad) + 1)
        if x0 >= x1 or y0 >= y1:
            continue
        xs = np.arange(x0, x1, dtype=np.float64)[None, :] - cx
        ys = np.arange(y0, y1, dtype=np.float64)[:, None] - cy
        q = (xs * xs + ys * ys) / (rad * rad)
        inside = q < 1.0
        region = lin[y0:y1, x0:x1]
        # spherical cap: nearer in the middle
        cap = region.min() * frac * (1.0 - 0.25 * np.sqrt(np.clip(1.0 - q, 0.0, 1.0)))
        lin[y0:y1, x0:x1] = np.where(inside, np.minimum(region, cap), region)
    # per-pixel integer hash dither, +-1e-4 relative
    xi = np.arange(width, dtype=np.uint64)[None, :]
    yi = np.arange(height, dtype=np.uint64)[:, None]
    hsh = (xi * np.uint64(73856093)) ^ (yi * np.uint64(19349663)) ^ np.uint64(seed & 0xFFFFFFFF)
    hsh = (hsh * np.uint64(2654435761)) & np.uint64(0xFFFFFFFF)
    dither = (hsh.astype(np.float64) / 4294967296.0 - 0.5) * 2.0e-4
    lin = np.clip(lin * (1.0 + dither), 0.011, 0.99)
    return linear01_to_raw(lin, cam)


_ATRIUM_CACHE: dict = {}


def _atrium_linear01(width: int, height: int, cam: Camera) -> np.ndarray:
    key = (width, height, cam)
    if key not in _ATRIUM_CACHE:
        _ATRIUM_CACHE.clear()                     # one scene at a time (a 4K f64 image is 66 MB)
        _ATRIUM_CACHE[key] = _atrium_linear01_uncached(width, height, cam)
    return _ATRIUM_CACHE[key].copy()


def _atrium_linear01_uncached(width: int, height: int, cam: Camera) -> np.ndarray:
    aspect = width / height
    th = math.tan(math.radians(cam.fov_y_deg) * 0.5)
    px = (np.arange(width, dtype=np.float64)[None, :] + 0.5) / width * 2.0 - 1.0
    py = 1.0 - (np.arange(height, dtype=np.float64)[:, None] + 0.5) / height * 2.0
    dx = np.broadcast_to(px * th * aspect, (height, width))
    dy = np.broadcast_to(py * th, (height, width))
    # camera at (0, 1.7, 0) looking down +z; t parameterises view-space z directly (dz = 1)
    eye_y = 1.7
    t = np.full((height, width), 60.0)                      # back wall at z = 60
    with np.errstate(divide="ignore", invalid="ignore"):
        tf = np.where(dy < 0, (0.0 - eye_y) / dy, np.inf)   # floor y = 0
        tc = np.where(dy > 0, (9.0 - eye_y) / dy, np.inf)   # ceiling y = 9
        tl = np.where(dx < 0, (-6.0) / dx, np.inf)          # walls x = -+6
        tr = np.where(dx > 0, (6.0) / dx, np.inf)
    for cand in (tf, tc, tl, tr):
        t = np.minimum(t, np.where(cand > 0, cand, np.inf))
    for row_x in (-3.5, 3.5):
        for i in range(8):
            cz, rad = 6.0 + 6.5 * i, 0.55
            # |(dx t - row_x, t - cz)| = rad   (vertical cylinder, any y)
            a = dx * dx + 1.0
            b = -2.0 * (dx * row_x + cz)
            c = row_x * row_x + cz * cz - rad * rad
            disc = b * b - 4.0 * a * c
            hit = disc > 0
            root = np.where(hit, (-b - np.sqrt(np.where(hit, disc, 0.0))) / (2.0 * a), np.inf)
            t = np.minimum(t, np.where(hit & (root > 0), root, np.inf))
    return np.clip(t / cam.far, cam.near / cam.far * 1.5, 0.99)


def atrium(width: int, height: int, cam: Camera = SPONZA_CAMERA) -> np.ndarray:
    """S3: analytic ray-cast of a floor, ceiling, two side walls, a back wall and two
    rows of 8 cylinders, seen through the Sponza scene camera."""
    return linear01_to_raw(_atrium_linear01(width, height, cam), cam)


def atrium_with_occluders(width: int, height: int, seed: int, n_boxes: int = 24, cam: Camera = SPONZA_CAMERA) -> np.ndarray:
    """S3 + seeded screen-space boxes standing in front of what they cover (S2's rectangle rule):
    distinct frames of the same scene for batches (bench.py: frame 0 of a batch is the plain atrium,
    frame f > 0 uses seed + f), so that a frame-index mix-up shows in the per-frame checksums."""
    lin = _atrium_linear01(width, height, cam)
    rng = _XorShift32(seed)
    scale = min(width, height)
    floor = cam.near / cam.far * 1.5
    for _ in range(n_boxes):
        cx, cy = rng.unit() * width, rng.unit() * height
        hw = (0.01 + 0.07 * rng.unit()) * scale
        hh = (0.02 + 0.12 * rng.unit()) * scale
        frac = 0.35 + 0.6 * rng.unit()
        x0, x1 = max(0, int(cx - hw)), min(width, int(cx + hw) + 1)
        y0, y1 = max(0, int(cy - hh)), min(height, int(cy + hh) + 1)
        if x0 >= x1 or y0 >= y1:
            continue
        region = lin[y0:y1, x0:x1]
        lin[y0:y1, x0:x1] = np.minimum(region, max(region.min() * frac, floor))
    return linear01_to_raw(lin, cam)


def make(kind: str, width: int, height: int, seed: int = 0x1234ABCD) -> np.ndarray:
    if kind in ("S1", "radial"):
        return radial_gradient(width, height)
    if kind in ("S2", "occluders"):
        return occluder_field(width, height, seed)
    if kind in ("S3", "atrium"):
        return atrium(width, height)
    raise ValueError(kind)
