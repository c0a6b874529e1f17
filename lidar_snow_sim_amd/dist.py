"""Frame-parallel sharding helpers (no data-path collective).

The reference's production loop (tools/snowfall/precompute.py:74-106) is a sequential loop over frames
with no carried state, so frames shard embarrassingly: rank r of W owns frames r, r + W, r + 2W, ...
Particle tables and laser constants are read-only and replicated.  The only cross-rank traffic is the
bench/driver bookkeeping below (a barrier and a max over ranks of the elapsed time), which runs over
torch.distributed -- RCCL ("nccl") on GPUs, gloo in the CPU tests.
"""
from __future__ import annotations

import os
from typing import List, Sequence


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard_indices(n_items: int, rank: int, world: int) -> List[int]:
    """Round-robin ownership: item i belongs to rank i % world (SURVEY 8 e)."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    return list(range(rank, n_items, world))


def bench_frame_seeds(rank: int, frames_per_rank: int, base: int = 1000) -> List[int]:
    """Weak scaling: every rank generates its own `frames_per_rank` synthetic sweeps; seeds never collide."""
    return [base + rank * frames_per_rank + f for f in range(frames_per_rank)]


def init(backend: str, rank: int, world: int, device=None):
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    kw = {}
    if device is not None:
        kw["device_id"] = device
    dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return dist


def max_over_ranks(value: float, device=None) -> float:
    """All-reduce(MAX) of a scalar; identity when not distributed."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(values: Sequence[float], device=None) -> List[float]:
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return [float(v) for v in values]
    t = torch.tensor(list(values), dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [float(v) for v in t.tolist()]
