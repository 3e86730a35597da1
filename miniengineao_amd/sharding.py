"""Frame sharding for multi-GPU batches.

The path shards across independent frames only: frame f goes to rank f mod world_size, each
rank owns a full context and all intermediates, and there is no data-path collective.
"""
from __future__ import annotations

from typing import List

import numpy as np


def frames_for_rank(num_frames: int, rank: int, world_size: int) -> List[int]:
    """Global frame indices processed by ``rank`` (round-robin, like the reference-free
    'one frame per GPU' of BASELINE config 4)."""
    if world_size < 1 or not (0 <= rank < world_size) or num_frames < 0:
        raise ValueError("bad (num_frames, rank, world_size)")
    return list(range(rank, num_frames, world_size))


def frame_seed(base_seed: int, frame: int) -> int:
    """Seed of synthetic frame ``frame`` (S2 uses seed + f for batch frame f)."""
    return (base_seed + frame) & 0xFFFFFFFF


def frame_checksum(arr: np.ndarray) -> int:
    """Order-sensitive 63-bit checksum of one output frame (FNV-style fold of 8-byte words): what
    the ranks exchange instead of frames when a multi-GPU batch is validated (bench.py --gpus N:
    one value per frame, gathered over RCCL, compared on rank 0)."""
    b = np.ascontiguousarray(arr).view(np.uint8).ravel()
    pad = (-len(b)) % 8
    if pad:
        b = np.concatenate([b, np.zeros(pad, np.uint8)])
    w = b.view(np.uint64)
    idx = np.arange(1, len(w) + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        mixed = (w ^ (idx * np.uint64(0x9E3779B97F4A7C15))) * np.uint64(0x100000001B3)
        return int(np.bitwise_xor.reduce(mixed) ^ np.uint64(len(b))) & 0x7FFFFFFFFFFFFFFF


def owner_of_frame(frame: int, world_size: int) -> int:
    """Rank that processes global frame ``frame`` (inverse of frames_for_rank)."""
    if world_size < 1 or frame < 0:
        raise ValueError("bad (frame, world_size)")
    return frame % world_size
