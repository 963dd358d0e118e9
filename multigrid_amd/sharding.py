"""Multi-GPU: shard the batch of independent envs over ranks; no data-path collective.

Envs never read each other's state (multigrid/base.py: one `MultiGridEnv` object shares nothing with another), so
the path is embarrassingly parallel: rank r of R owns the contiguous env range
[first, first + count) of the global batch, with its own state tensors on its own GPU.  Per-env seeds and synthetic
RNG states are functions of the GLOBAL env index, so the concatenation of the shards' outputs is bit-identical
to a single-device run of the whole batch, for any R.  `torch.distributed` (RCCL on ROCm) is only ever used by
callers for barriers / timing reductions / optional gathers of results -- never inside `step`.
"""
from __future__ import annotations

import os

import torch

from .batched import BatchedMultiGridEnv
from .spec import EnvSpec


def shard_range(global_batch: int, rank: int, world_size: int) -> tuple[int, int]:
    """Contiguous block partition: (first_env, count).  The first `global_batch % world_size` ranks get one extra."""
    if not 0 <= rank < world_size:
        raise ValueError(f"rank {rank} outside world of {world_size}")
    base, extra = divmod(global_batch, world_size)
    count = base + (1 if rank < extra else 0)
    first = rank * base + min(rank, extra)
    return first, count


def rank_world_from_env() -> tuple[int, int, int]:
    """(rank, local_rank, world_size) as set by `python -m torch.distributed.run`."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def make_sharded_env(spec: EnvSpec, global_batch: int, rank: int, world_size: int, device=None,
                     backend=None) -> BatchedMultiGridEnv:
    """This rank's shard of a `global_batch`-env job."""
    first, count = shard_range(global_batch, rank, world_size)
    if device is None:
        device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", rank)))
    return BatchedMultiGridEnv(spec, count, device, first_env=first, backend=backend)
