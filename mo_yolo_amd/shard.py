"""Multi-GPU sharding of the tracking path: independent sequences, one process per GPU, no
collective on the data path (the reference's state is per MOTRTrack instance,
head.py:121,141).  torch.distributed is used only to agree on timing / gather results."""
from __future__ import annotations

from typing import List

import torch
import torch.distributed as dist


def sequences_for_rank(n_sequences: int, rank: int, world: int) -> List[int]:
    """Sequence i runs on rank i mod world."""
    return [i for i in range(n_sequences) if i % world == rank]


def max_over_ranks(seconds: float, device=None) -> float:
    """Wall time of the slowest rank (the bench contract's MAX over ranks)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def whole_job_fps(frames_per_rank: int, seconds_max: float, world: int) -> float:
    return frames_per_rank * world / seconds_max


def gather_objects(obj):
    """Host-side collection of per-rank results (txt rows, HOTA scalars) on every rank."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [obj]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, obj)
    return out
