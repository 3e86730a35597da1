"""miniengineao_amd -- MI355X-native multi-scale SSAO (hot path of keijiro/MiniEngineAO).

The product is ``lib/libmeao_hip.so`` (hand-written gfx950 kernels behind the C ABI of
``include/meao.h``).  This package is the thin host side: the ctypes binding, the
``AmbientOcclusion`` mirror of the reference component, synthetic depth inputs and the
frame-sharding helper for multi-GPU batches.  Importing it does not load the library;
using the hot path without the built library raises ImportError (no CPU fallback).
"""
from . import synth  # noqa: F401
from .sharding import frames_for_rank  # noqa: F401

__all__ = ["AmbientOcclusion", "AmbientOcclusionPool", "synth", "frames_for_rank"]


def __getattr__(name):
    if name in ("AmbientOcclusion", "AmbientOcclusionPool"):
        from . import ambient_occlusion
        return getattr(ambient_occlusion, name)
    raise AttributeError(name)
