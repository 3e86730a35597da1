"""Multi-GPU plumbing for the batch case: one process per GPU, frames sharded across ranks,
no data-path collective.  torch.distributed is used only for the barrier, the max-over-ranks of
a timed region and gathering per-frame checksums (backend "nccl" is RCCL on ROCm; "gloo" on
CPU for tests)."""
from __future__ import annotations

import os
from typing import List, Tuple

import torch
import torch.distributed as dist


def env_world() -> Tuple[int, int, int]:
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def free_port() -> int:
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


_free_port = free_port


def init(backend: str = "nccl", device: "torch.device | None" = None) -> Tuple[int, int, int]:
    """Initialise the default process group when WORLD_SIZE > 1.  Returns (rank, world, local_rank)."""
    rank, world, local_rank = env_world()
    # one rank still gets a group when a launcher started it (torch.distributed.run exports WORLD_SIZE; `bench.py --gpus 1
    # --launcher`) or on request: the collective path -- RCCL communicator, barrier, all_reduce, all_gather -- with a single rank
    force = os.environ.get("MEAO_FORCE_DIST") == "1" or "TORCHELASTIC_RUN_ID" in os.environ
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            # torch.distributed.run always exports MASTER_PORT; only a hand-made single-rank group
            # (MEAO_FORCE_DIST=1) gets here, and it takes a free port instead of a hard-wired one
            if world > 1:
                raise RuntimeError("MASTER_PORT is not set: launch with torch.distributed.run (or export MASTER_ADDR / MASTER_PORT)")
            os.environ["MASTER_PORT"] = str(free_port())
        kwargs = {}
        if backend == "nccl" and device is not None:
            kwargs["device_id"] = device
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kwargs)
    return rank, world, local_rank


def _tensor_device(device):
    if dist.is_initialized() and dist.get_backend() == "nccl":
        return device
    return torch.device("cpu")


def fence(device: "torch.device | None" = None) -> None:
    """synchronize + barrier + synchronize: both sides of a timed region."""
    if device is not None and device.type == "cuda":
        torch.cuda.synchronize(device)
    if dist.is_initialized():
        dist.barrier()
        if device is not None and device.type == "cuda":
            torch.cuda.synchronize(device)


def max_over_ranks(value: float, device: "torch.device | None" = None) -> float:
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=_tensor_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def group_info() -> dict:
    """What the collectives of this process go through: nothing (single process, no group), or a group of `backend`."""
    if not dist.is_initialized():
        return {"initialized": False, "backend": None, "world": 1}
    return {"initialized": True, "backend": str(dist.get_backend()), "world": dist.get_world_size()}


def world_size() -> int:
    """World size as the process group reports it (1 without a group)."""
    return dist.get_world_size() if dist.is_initialized() else 1


def gather_floats(value: float, device: "torch.device | None" = None) -> List[float]:
    """One float per rank, indexed by rank (e.g. each rank's own ms per step)."""
    if not dist.is_initialized():
        return [float(value)]
    t = torch.tensor([value], dtype=torch.float64, device=_tensor_device(device))
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(x.item()) for x in out]


def gather_checksums(local: List[int], device: "torch.device | None" = None) -> List[List[int]]:
    """All ranks' per-frame 63-bit checksums, indexed [rank][local frame]."""
    if not dist.is_initialized():
        return [list(local)]
    world = dist.get_world_size()
    n = torch.tensor([len(local)], dtype=torch.int64, device=_tensor_device(device))
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n)
    width = max(int(c.item()) for c in counts)
    mine = torch.zeros(max(width, 1), dtype=torch.int64, device=_tensor_device(device))
    if local:
        mine[: len(local)] = torch.tensor([v & 0x7FFFFFFFFFFFFFFF for v in local], dtype=torch.int64)
    out = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(out, mine)
    return [[int(v) for v in out[r][: int(counts[r].item())].tolist()] for r in range(world)]


def shutdown() -> None:
    if dist.is_initialized():
        dist.destroy_process_group()
