"""Frame sharding for multi-GPU batches.

The path shards across independent frames only: frame f goes to rank f mod world_size, each
rank owns a full context and all intermediates, and there is no data-path collective.
"""
from __future__ import annotations

from typing import List


def frames_for_rank(num_frames: int, rank: int, world_size: int) -> List[int]:
    """Global frame indices processed by ``rank`` (round-robin, like the reference-free
    'one frame per GPU' of BASELINE config 4)."""
    if world_size < 1 or not (0 <= rank < world_size) or num_frames < 0:
        raise ValueError("bad (num_frames, rank, world_size)")
    return list(range(rank, num_frames, world_size))


def frame_seed(base_seed: int, frame: int) -> int:
    """Seed of synthetic frame ``frame`` (S2 uses seed + f for batch frame f)."""
    return (base_seed + frame) & 0xFFFFFFFF
