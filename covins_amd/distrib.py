"""Multi-GPU plumbing of bench.py: one process per GPU, `torch.distributed` (backend "nccl" = RCCL on ROCm; "gloo"
in the CPU tests). Round 1 shards by MAP: rank r optimises its own merged map, there is no
data-path collective; the only collectives are the barrier around the timed region, the MAX of the wall time and
the SUM of the executed iterations."""
from __future__ import annotations

import os
from typing import Optional, Tuple


def env_ranks() -> Tuple[int, int, int]:
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init(backend: str = "nccl", local_rank: int = 0):
    """Returns the torch.distributed module (initialised) or None for a single process."""
    _, _, world = env_ranks()
    if world <= 1:
        return None
    import torch
    import torch.distributed as dist
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group(backend=backend)
    return dist


def map_seed_for_rank(rank: int, base_seed: int = 0) -> int:
    """Map-sharded weak scaling: every rank owns a differently seeded merged map of the same configuration."""
    return base_seed + rank


def barrier(dist, device: Optional[str] = None) -> None:
    import torch
    if device is not None and device.startswith("cuda"):
        torch.cuda.synchronize(device)
    if dist is not None:
        dist.barrier()
        if device is not None and device.startswith("cuda"):
            torch.cuda.synchronize(device)


def aggregate(dt: float, iterations: float, dist, device: str = "cpu") -> Tuple[float, float]:
    """(max over ranks of the wall time, sum over ranks of the executed trust-region iterations)."""
    if dist is None:
        return float(dt), float(iterations)
    import torch
    t = torch.tensor([dt], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    it = torch.tensor([float(iterations)], dtype=torch.float64, device=device)
    dist.all_reduce(it, op=dist.ReduceOp.SUM)
    return float(t.item()), float(it.item())


def throughput(dt_max: float, iterations_sum: float) -> float:
    """Whole-job GBA iterations per second."""
    return iterations_sum / dt_max
