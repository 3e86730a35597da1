"""Host-side mirror of the reference component ``MiniEngineAO.AmbientOcclusion``
(Assets/MiniEngineAO/AmbientOcclusion.cs, "AO.cs") over the C ABI of libmeao_hip.so.

The reference is a Unity MonoBehaviour: six public properties (AO.cs:22-66), implicit camera
inputs (AO.cs:339-340,563-573), depth texture in, "AmbientOcclusion" R8 texture out
(AO.cs:475,824).  This class keeps the property names and meaning; what Unity did implicitly
(camera, depth texture, render-texture output) is explicit here.  Property changes are picked
up lazily before the next frame, like CheckPropertiesChanged (AO.cs:104-113).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np

from . import _lib as L

_NP_OF_FMT = {L.FMT_F32: np.float32, L.FMT_F16: np.uint16, L.FMT_UNORM8: np.uint8}

# _debug values (AO.cs:789-808)
DEBUG_BUFFER_NAMES = {
    1: "LinearDepth", 2: "LowDepth1", 3: "LowDepth2", 4: "LowDepth3", 5: "LowDepth4",
    6: "TiledDepth1", 7: "TiledDepth2", 8: "TiledDepth3", 9: "TiledDepth4",
    10: "Occlusion1", 11: "Occlusion2", 12: "Occlusion3", 13: "Occlusion4",
    14: "Combined1", 15: "Combined2", 16: "Combined3", 17: "AmbientOcclusion",
}
# beyond the reference's list: the Render.main (wide) targets of the hq_levels variant
HQ_BUFFER_NAMES = {18: "OcclusionHQ1", 19: "OcclusionHQ2", 20: "OcclusionHQ3", 21: "OcclusionHQ4"}


class AmbientOcclusion:
    """depth in -> AO texture out, on one MI355X.  One instance per device."""

    def __init__(self, width: int, height: int, *, device: int = 0, num_levels: int = 4,
                 ao_format: int = L.AO_R8, f16_rounding: int = L.F16_RTZ_CLAMP,
                 max_batch: int = 1, depth_format: int = L.DEPTH_F32,
                 near_clip: float = 0.3, far_clip: float = 1000.0,
                 projection00: Optional[float] = None, reversed_z: bool = True,
                 hq_levels: int = 0, sample_set: int = L.SAMPLES_CHECKER, single_pass_stereo: bool = False,
                 pipelined: bool = False):
        """hq_levels / sample_set / single_pass_stereo: variants the reference's shaders and host carry
        but its command buffer never (or only in VR) uses; see include/meao.h.  ``width`` is the
        double-wide eye pair when single_pass_stereo is set (AO.cs:339)."""
        self._lib = L.load()
        cfg = L.Config()
        self._lib.meao_default_config(C.byref(cfg))
        cfg.device, cfg.width, cfg.height = device, width, height
        cfg.num_levels, cfg.ao_format, cfg.f16_rounding = num_levels, ao_format, f16_rounding
        cfg.max_batch = max_batch
        cfg.depth_format = depth_format
        cfg.hq_levels, cfg.sample_set = hq_levels, sample_set
        cfg.pipelined = 1 if pipelined else 0   # second downsample set from the start (prefetch_device never allocates)
        self._cfg = cfg
        prm = L.Params()
        self._lib.meao_default_params(C.byref(prm))
        prm.near_clip, prm.far_clip = near_clip, far_clip
        if projection00 is not None:
            prm.proj00 = projection00
        prm.reversed_z = 1 if reversed_z else 0
        prm.single_pass_stereo = 1 if single_pass_stereo else 0
        self._prm = prm
        self._ambient_only = True          # AO.cs:68; composite-only flag, kept for surface parity
        self._debug = 0                    # AO.cs:60
        self._dirty = True
        self._ctx = C.c_void_p()
        L.check(self._lib.meao_create(C.byref(cfg), C.byref(self._ctx)))

    # ---- the six public properties of the reference (AO.cs:22-66) ----------------------
    def _get(self, name):
        return getattr(self._prm, name)

    def _set(self, name, value):
        if getattr(self._prm, name) != value:   # CheckUpdate (AO.cs:91-102)
            setattr(self._prm, name, value)
            self._dirty = True

    noiseFilterTolerance = property(lambda s: s._get("noise_filter_tolerance"),
                                    lambda s, v: s._set("noise_filter_tolerance", float(v)))
    blurTolerance = property(lambda s: s._get("blur_tolerance"),
                             lambda s, v: s._set("blur_tolerance", float(v)))
    upsampleTolerance = property(lambda s: s._get("upsample_tolerance"),
                                 lambda s, v: s._set("upsample_tolerance", float(v)))
    thicknessModifier = property(lambda s: s._get("thickness_modifier"),
                                 lambda s, v: s._set("thickness_modifier", float(v)))
    intensity = property(lambda s: s._get("intensity"), lambda s, v: s._set("intensity", float(v)))

    @property
    def ambientOnly(self) -> bool:
        return self._ambient_only

    @ambientOnly.setter
    def ambientOnly(self, value: bool) -> None:
        self._ambient_only = bool(value)

    # ---- camera terms Unity supplied implicitly (AO.cs:563-573) ------------------------
    nearClipPlane = property(lambda s: s._get("near_clip"), lambda s, v: s._set("near_clip", float(v)))
    farClipPlane = property(lambda s: s._get("far_clip"), lambda s, v: s._set("far_clip", float(v)))
    projection00 = property(lambda s: s._get("proj00"), lambda s, v: s._set("proj00", float(v)))
    usesReversedZBuffer = property(lambda s: bool(s._get("reversed_z")),
                                   lambda s, v: s._set("reversed_z", 1 if v else 0))
    singlePassStereoEnabled = property(lambda s: bool(s._get("single_pass_stereo")),       # AO.cs:392-401
                                       lambda s, v: s._set("single_pass_stereo", 1 if v else 0))

    # ---- geometry ----------------------------------------------------------------------
    width = property(lambda s: s._cfg.width)
    height = property(lambda s: s._cfg.height)
    ao_format = property(lambda s: s._cfg.ao_format)
    max_batch = property(lambda s: s._cfg.max_batch)
    ao_dtype = property(lambda s: np.uint8 if s._cfg.ao_format == L.AO_R8 else np.uint16)

    def resize(self, width: int, height: int) -> None:
        """Screen-size change (AO.cs:338-341)."""
        L.check(self._lib.meao_resize(self._ctx, width, height), self._ctx)
        self._cfg.width, self._cfg.height = width, height

    def _sync_params(self) -> None:
        if self._dirty:
            L.check(self._lib.meao_set_params(self._ctx, C.byref(self._prm)), self._ctx)
            self._dirty = False

    # ---- the hot path ------------------------------------------------------------------
    def render(self, depth: np.ndarray) -> np.ndarray:
        """One frame, host arrays: (H, W) raw depth in the configured depth_format (float32 by
        default; uint16 / uint32 codes for UNORM16 / UNORM24 / F16 bits) -> AO (H, W) uint8 / f16 bits."""
        return self.render_batch([depth])[0]

    def render_batch(self, depths: Sequence[np.ndarray]) -> list:
        n = len(depths)
        self._sync_params()
        dt = {L.DEPTH_F32: np.float32, L.DEPTH_UNORM16: np.uint16, L.DEPTH_UNORM24: np.uint32,
              L.DEPTH_F16: np.uint16}[self._cfg.depth_format]
        ins = [np.ascontiguousarray(d, dtype=dt) for d in depths]
        for d in ins:
            if d.shape != (self.height, self.width):
                raise ValueError(f"depth shape {d.shape} != ({self.height}, {self.width})")
        outs = [np.empty((self.height, self.width), self.ao_dtype) for _ in range(n)]
        pin = (C.c_void_p * n)(*[d.ctypes.data for d in ins])
        pout = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
        L.check(self._lib.meao_execute_batch(self._ctx, n, pin, L.MEM_HOST, pout, L.MEM_HOST, None), self._ctx)
        return outs

    def execute_device(self, depth_ptrs: Sequence[int], out_ptrs: Sequence[int], stream: int = 0) -> None:
        """Device-resident frames (raw device addresses, e.g. torch ``data_ptr()``); asynchronous."""
        n = len(depth_ptrs)
        self._sync_params()
        pin = (C.c_void_p * n)(*depth_ptrs)
        pout = (C.c_void_p * n)(*out_ptrs)
        L.check(self._lib.meao_execute_batch(self._ctx, n, pin, L.MEM_DEVICE, pout, L.MEM_DEVICE,
                                             C.c_void_p(stream) if stream else None), self._ctx)

    def prefetch_device(self, depth_ptrs: Sequence[int]) -> None:
        """Announce the device depth frames of the call after next (meao_prefetch_batch): the next
        execute_device() carries their downsample pass inside its last upsample kernel."""
        n = len(depth_ptrs)
        self._sync_params()     # pending property changes first: meao_set_params would drop the announcement
        pin = (C.c_void_p * n)(*depth_ptrs)
        L.check(self._lib.meao_prefetch_batch(self._ctx, n, pin), self._ctx)

    def synchronize(self, stream: int = 0) -> None:
        L.check(self._lib.meao_synchronize(self._ctx, C.c_void_p(stream) if stream else None), self._ctx)

    # ---- composite (PushCompositeCommands, AO.cs:822-839) ------------------------------
    def composite(self, ao: np.ndarray, color_rgba16f: np.ndarray, gbuffer0_rgba8: Optional[np.ndarray] = None,
                  debug: bool = False) -> None:
        """Host arrays, in place.  Mode follows the reference: debug view if ``debug`` (AO.cs:826),
        ambient-only into GBuffer0 + the HDR target if ``ambientOnly`` and a GBuffer0 is given
        (AO.cs:830-834), else the standard multiply (AO.cs:837)."""
        assert color_rgba16f.dtype == np.uint16 and color_rgba16f.shape == (self.height, self.width, 4)
        assert ao.dtype == self.ao_dtype and ao.shape == (self.height, self.width)
        if debug:
            mode, g = L.COMPOSITE_DEBUG, None
        elif self._ambient_only and gbuffer0_rgba8 is not None:
            assert gbuffer0_rgba8.dtype == np.uint8 and gbuffer0_rgba8.shape == (self.height, self.width, 4)
            mode, g = L.COMPOSITE_AMBIENT_ONLY, gbuffer0_rgba8.ctypes.data
        else:
            mode, g = L.COMPOSITE_MULTIPLY, None
        L.check(self._lib.meao_composite(self._ctx, mode, ao.ctypes.data, color_rgba16f.ctypes.data, g,
                                         L.MEM_HOST, None), self._ctx)

    def composite_device(self, mode: int, ao_ptr: int, color_ptr: int, gbuffer0_ptr: int = 0, stream: int = 0) -> None:
        L.check(self._lib.meao_composite(self._ctx, mode, ao_ptr, color_ptr, gbuffer0_ptr or None, L.MEM_DEVICE,
                                         C.c_void_p(stream) if stream else None), self._ctx)

    def composite_enqueue_device(self, mode: int, ao_ptrs: Sequence[int], color_ptrs: Sequence[int],
                                 gbuffer0_ptrs: Optional[Sequence[int]] = None) -> None:
        """Composite of device frames an earlier execute produced; rides inside the NEXT execute's render
        kernel (meao_composite_enqueue).  composite_flush() runs whatever still waits."""
        n = len(ao_ptrs)
        g = (C.c_void_p * n)(*gbuffer0_ptrs) if gbuffer0_ptrs else None
        L.check(self._lib.meao_composite_enqueue(self._ctx, mode, n, (C.c_void_p * n)(*ao_ptrs),
                                                 (C.c_void_p * n)(*color_ptrs), g), self._ctx)

    def composite_flush(self, stream: int = 0) -> None:
        L.check(self._lib.meao_composite_flush(self._ctx, C.c_void_p(stream) if stream else None), self._ctx)

    @property
    def composite_pending(self) -> bool:
        """A batch enqueued with composite_enqueue_device() that no execute / flush / resize has run yet
        (meao_composite_pending: the library's own state, not a mirror of it)."""
        if not self._ctx:
            return False
        n = C.c_int32()
        L.check(self._lib.meao_composite_pending(self._ctx, C.byref(n)), self._ctx)
        return n.value > 0

    # ---- observability (the _debug views, AO.cs:787-820) -------------------------------
    def debug_buffer(self, debug_id: int, frame: int = 0) -> np.ndarray:
        d = L.Desc()
        L.check(self._lib.meao_get_intermediate(self._ctx, frame, debug_id, None, 0, L.MEM_HOST, C.byref(d)), self._ctx)
        shape = (d.slices, d.height, d.width) if d.slices > 1 else (d.height, d.width)
        out = np.empty(shape, _NP_OF_FMT[d.format])
        L.check(self._lib.meao_get_intermediate(self._ctx, frame, debug_id, out.ctypes.data, out.nbytes,
                                                L.MEM_HOST, C.byref(d)), self._ctx)
        return out

    def debug_view(self, debug_id: int, frame: int = 0) -> np.ndarray:
        """What `_debug = debug_id` shows (AO.cs:787-820): the buffer blitted into the AO target."""
        out = np.empty((self.height, self.width), self.ao_dtype)
        L.check(self._lib.meao_debug_view(self._ctx, frame, debug_id, out.ctypes.data, L.MEM_HOST, None), self._ctx)
        return out

    def set_profiling(self, enable) -> None:
        """False / True, or an int N > 1: HIP events around the passes of every Nth execute only (meao_set_profiling)."""
        L.check(self._lib.meao_set_profiling(self._ctx, int(enable)), self._ctx)

    def hostile_frames(self) -> int:
        """Bit mask of the frames of the last execute that ran the IEEE-division bodies (meao_hostile_frames)."""
        m = C.c_uint64()
        L.check(self._lib.meao_hostile_frames(self._ctx, C.byref(m)), self._ctx)
        return m.value

    def debug_set(self, key: int, value: int) -> None:
        """meao_debug_set: launch-structure overrides (identical results)."""
        L.check(self._lib.meao_debug_set(self._ctx, key, value), self._ctx)

    def set_tracing(self, enable: bool) -> None:
        """roctx ranges around every pass (rocprofv3 --marker-trace)."""
        L.check(self._lib.meao_set_tracing(self._ctx, 1 if enable else 0), self._ctx)

    def pass_times_ms(self):
        ms = (C.c_float * L.NUM_PASSES)()
        n = C.c_int32()
        L.check(self._lib.meao_get_pass_times(self._ctx, C.byref(ms), C.byref(n)), self._ctx)
        return list(ms), n.value

    def algorithmic_bytes(self):
        b = (C.c_uint64 * L.NUM_PASSES)()
        L.check(self._lib.meao_algorithmic_bytes(C.byref(self._cfg), C.byref(b)))
        return list(b)

    def selftest(self, which: int) -> int:
        n = C.c_uint64()
        L.check(self._lib.meao_selftest(self._ctx, which, C.byref(n)), self._ctx)
        return n.value

    def close(self, flush_composite: bool = False) -> None:
        """meao_destroy.  A composite batch still waiting is DISCARDED by the library (its targets are caller memory
        that is usually gone when a context dies): pass flush_composite=True while those buffers are alive to run it
        first; otherwise the loss is reported as a RuntimeWarning instead of passing silently."""
        if self._ctx:
            if self.composite_pending:
                if flush_composite:
                    self.composite_flush()
                else:
                    import warnings
                    warnings.warn("AmbientOcclusion.close(): a composite batch was still waiting and is discarded "
                                  "(call composite_flush() or close(flush_composite=True) first)", RuntimeWarning, stacklevel=2)
            self._lib.meao_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class AmbientOcclusionPool:
    """One context per device behind the C ABI's meao_pool_*: frame f of a batch runs on member
    f mod G, one host thread, no data-path exchange.  ``devices`` may repeat an ordinal
    (several members on one GPU)."""

    def __init__(self, width: int, height: int, devices: Sequence[int], *, max_batch: int = 1,
                 ao_format: int = L.AO_R8, near_clip: float = 0.3, far_clip: float = 1000.0,
                 projection00: Optional[float] = None, reversed_z: bool = True, intensity: float = 1.0,
                 pipelined: bool = False):
        self._lib = L.load()
        cfg = L.Config()
        self._lib.meao_default_config(C.byref(cfg))
        cfg.width, cfg.height, cfg.max_batch, cfg.ao_format = width, height, max_batch, ao_format
        cfg.pipelined = 1 if pipelined else 0
        self._cfg = cfg
        self.devices = list(devices)
        self._pool = C.c_void_p()
        arr = (C.c_int32 * len(self.devices))(*self.devices)
        status = self._lib.meao_pool_create(C.byref(cfg), arr, len(self.devices), C.byref(self._pool))
        if status != L.OK:
            raise L.MeaoError(status, self._lib.meao_pool_last_error(None).decode())
        prm = L.Params()
        self._lib.meao_default_params(C.byref(prm))
        prm.near_clip, prm.far_clip, prm.reversed_z, prm.intensity = near_clip, far_clip, 1 if reversed_z else 0, intensity
        if projection00 is not None:
            prm.proj00 = projection00
        try:
            self._check(self._lib.meao_pool_set_params(self._pool, C.byref(prm)))
        except Exception:
            self.close()        # the pool exists already: do not leak its contexts
            raise

    def _check(self, status: int) -> None:
        if status != L.OK:
            raise L.MeaoError(status, self._lib.meao_pool_last_error(self._pool).decode())

    size = property(lambda s: s._lib.meao_pool_size(s._pool))

    def device_of_frame(self, frame: int) -> int:
        return self._lib.meao_pool_device_of_frame(self._pool, frame)

    def render_batch(self, depths: Sequence[np.ndarray]) -> list:
        """Host arrays in and out."""
        n = len(depths)
        ins = [np.ascontiguousarray(d, dtype=np.float32) for d in depths]
        dt = np.uint8 if self._cfg.ao_format == L.AO_R8 else np.uint16
        outs = [np.empty((self._cfg.height, self._cfg.width), dt) for _ in range(n)]
        pin = (C.c_void_p * n)(*[d.ctypes.data for d in ins])
        pout = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
        self._check(self._lib.meao_pool_execute_batch(self._pool, n, pin, L.MEM_HOST, pout, L.MEM_HOST))
        return outs

    def execute_device(self, depth_ptrs: Sequence[int], out_ptrs: Sequence[int]) -> None:
        """Frame f resident on device_of_frame(f); asynchronous (synchronize())."""
        n = len(depth_ptrs)
        pin, pout = (C.c_void_p * n)(*depth_ptrs), (C.c_void_p * n)(*out_ptrs)
        self._check(self._lib.meao_pool_execute_batch(self._pool, n, pin, L.MEM_DEVICE, pout, L.MEM_DEVICE))

    def prefetch_device(self, depth_ptrs: Sequence[int]) -> None:
        """meao_pool_prefetch_batch: the frames of the call after next, dealt like execute_device deals them."""
        n = len(depth_ptrs)
        self._check(self._lib.meao_pool_prefetch_batch(self._pool, n, (C.c_void_p * n)(*depth_ptrs)))

    def composite_enqueue_device(self, mode: int, ao_ptrs: Sequence[int], color_ptrs: Sequence[int],
                                 gbuffer0_ptrs: Optional[Sequence[int]] = None) -> None:
        n = len(ao_ptrs)
        g = (C.c_void_p * n)(*gbuffer0_ptrs) if gbuffer0_ptrs else None
        self._check(self._lib.meao_pool_composite_enqueue(self._pool, mode, n, (C.c_void_p * n)(*ao_ptrs),
                                                          (C.c_void_p * n)(*color_ptrs), g))

    def composite_flush(self) -> None:
        self._check(self._lib.meao_pool_composite_flush(self._pool))

    @property
    def composite_pending(self) -> bool:
        """Frames of an enqueued composite batch still waiting in any member (meao_pool_composite_pending)."""
        if not self._pool:
            return False
        n = C.c_int32()
        self._check(self._lib.meao_pool_composite_pending(self._pool, C.byref(n)))
        return n.value > 0

    def configure(self, key: int, value: int) -> None:
        """meao_pool_configure: L.POOL_SPIN_US (how long a worker spins for its next job), L.POOL_BIND_NUMA (workers bind to their
        device's NUMA node; before the first DEVICE batch)."""
        self._check(self._lib.meao_pool_configure(self._pool, key, value))

    def member_placement(self, member: int) -> dict:
        """{"numa_node": node of the member's device (-1 unknown), "worker_bound": its worker thread runs on that node}."""
        node, bound = C.c_int32(), C.c_int32()
        self._check(self._lib.meao_pool_member_placement(self._pool, member, C.byref(node), C.byref(bound)))
        return {"numa_node": node.value, "worker_bound": bool(bound.value)}

    def gather_path(self, member: int, dst_device: int) -> int:
        """L.POOL_PATH_*: how gather_to_device copies from `member`'s device to dst_device."""
        return self._lib.meao_pool_gather_path(self._pool, member, dst_device)

    def member_context(self, member: int):
        """Raw meao_ctx* of a member (owned by the pool), e.g. for meao_set_profiling / meao_get_pass_times."""
        return C.c_void_p(self._lib.meao_pool_context(self._pool, member))

    def gather_to_device(self, src_ptrs: Sequence[int], dst_ptrs: Sequence[int], dst_device: int) -> None:
        n = len(src_ptrs)
        self._check(self._lib.meao_pool_gather_to_device(self._pool, n, (C.c_void_p * n)(*src_ptrs),
                                                         (C.c_void_p * n)(*dst_ptrs), dst_device))

    def synchronize(self) -> None:
        self._check(self._lib.meao_pool_synchronize(self._pool))

    def close(self, flush_composite: bool = False) -> None:
        """meao_pool_destroy; a composite still waiting is discarded unless flush_composite (see AmbientOcclusion.close)."""
        if self._pool:
            if self.composite_pending:
                if flush_composite:
                    self.composite_flush()
                else:
                    import warnings
                    warnings.warn("AmbientOcclusionPool.close(): a composite batch was still waiting and is discarded "
                                  "(call composite_flush() or close(flush_composite=True) first)", RuntimeWarning, stacklevel=2)
            self._lib.meao_pool_destroy(self._pool)
            self._pool = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
