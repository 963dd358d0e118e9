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


class NodeEnv:
    """ONE process driving every GPU of a node (round 6): the shards `make_sharded_env` gives the ranks of a
    `torch.distributed.run` job, all held by this process -- shard r of the global batch on `devices[r]`, same partition, same
    global-index seeds and layout choice, hence the same results as R processes or one device, bit for bit.  No collective, no
    threads: a launch (or a hipGraph replay) is asynchronous, so one Python thread that issues shard 0's, shard 1's, ... leaves all
    devices working at once.  The per-call host cost (a few microseconds per device) bounds an EAGER loop at the small per-GPU
    batches of a strong split; `capture_steps` puts T steps of every shard into a graph per device, and `NodeGraph.replay()` then
    costs one replay call per device for all T steps -- the form a single learner process uses to drive the node.

    `devices`: one entry per shard (torch devices or strings; the same device may appear more than once: the one-GPU test).
    `backend_factory(spec, device)`: launcher per shard (tests: the CPU oracle backend)."""

    def __init__(self, spec: EnvSpec, global_batch: int, devices, backend_factory=None):
        self.spec, self.global_batch = spec, int(global_batch)
        self.devices = [torch.device(d) for d in devices]
        R = len(self.devices)
        self.shards: list[BatchedMultiGridEnv] = []
        self.ranges = []
        for r, dev in enumerate(self.devices):
            first, count = shard_range(self.global_batch, r, R)
            be = backend_factory(spec, dev) if backend_factory is not None else None
            self.shards.append(BatchedMultiGridEnv(spec, count, dev, first_env=first, backend=be))
            self.ranges.append((first, first + count))

    def load_state(self, grid, agents, rng=None, aux=None, step_count=None, validate: bool = True):
        """The global batch's state (leading dimension `global_batch`, or one env's state to broadcast), cut per shard."""
        import numpy as np
        def cut(x, lo, hi):
            if x is None:
                return None
            a = x.cpu().numpy() if torch.is_tensor(x) else np.asarray(x)
            return a[lo:hi] if a.ndim and a.shape[0] == self.global_batch and self.global_batch > 1 else a
        for sh, (lo, hi) in zip(self.shards, self.ranges):
            sh.load_state(cut(grid, lo, hi), cut(agents, lo, hi), cut(rng, lo, hi), cut(aux, lo, hi), cut(step_count, lo, hi),
                          validate=validate)

    def seed(self, seed: int):
        for sh in self.shards:
            sh.seed(seed)

    def seed_synthetic(self, seed: int):
        for sh in self.shards:
            sh.seed_synthetic(seed)

    def set_layout_pool(self, grids, agents, auxs=None):
        for sh in self.shards:
            sh.set_layout_pool(grids, agents, auxs)

    def scatter(self, x) -> list:
        """A tensor over the global batch (dimension 0, or dimension 1 of a [T, B, ...] action script when `x.dim() >= 3`) -> its
        per-shard pieces on the shards' devices."""
        t = x if torch.is_tensor(x) else torch.as_tensor(x)
        d = 1 if t.dim() >= 3 else 0
        return [t.narrow(d, lo, hi - lo).contiguous().to(dev) for (lo, hi), dev in zip(self.ranges, self.devices)]

    def step(self, actions, auto_reset: bool = False, one_hot: bool = False) -> list:
        """One step of every shard (issued shard after shard, all asynchronous).  `actions`: the per-shard list `scatter` makes, or
        one tensor over the global batch.  Returns the shards' output tuples."""
        parts = actions if isinstance(actions, (list, tuple)) else self.scatter(actions)
        return [sh.step(a, auto_reset=auto_reset, one_hot=one_hot) for sh, a in zip(self.shards, parts)]

    def capture_steps(self, actions, auto_reset: bool = False, one_hot: bool = False, sub_shards=1) -> "NodeGraph":
        """A hipGraph of `T` steps per shard (actions: per-shard [T, b_r, A] tensors, or one [T, B, A] script to scatter; kept by
        reference as in `BatchedMultiGridEnv.capture_steps`)."""
        parts = actions if isinstance(actions, (list, tuple)) else self.scatter(actions)
        graphs = []
        for sh, a in zip(self.shards, parts):
            if sh.device.type == "cuda":
                with torch.cuda.device(sh.device):
                    graphs.append(sh.capture_steps(a, auto_reset=auto_reset, one_hot=one_hot, sub_shards=sub_shards))
            else:                                   # (host launchers -- the test-suite's oracle backend -- have no graphs: eager replay)
                graphs.append(_EagerReplay(sh, a, auto_reset, one_hot))
        return NodeGraph(self, graphs, parts)

    def synchronize(self):
        for sh in self.shards:
            sh.join()
            if sh.device.type == "cuda":
                torch.cuda.current_stream(sh.device).synchronize()

    def gather(self, name: str) -> torch.Tensor:
        """Attribute `name` of every shard (`obs`, `reward`, `grid`, `agents`, `rng`, ...) concatenated on the host, in env order."""
        self.synchronize()
        return torch.cat([getattr(sh, name).cpu() for sh in self.shards])

    def check_errors(self):
        for sh in self.shards:
            sh.check_errors()


class _EagerReplay:
    def __init__(self, env, actions, auto_reset, one_hot):
        self.env, self.actions, self.kw = env, actions, dict(auto_reset=auto_reset, one_hot=one_hot)

    def replay(self):
        for t in range(self.actions.shape[0]):
            self.env.step(self.actions[t], **self.kw)


class NodeGraph:
    """The per-device graphs of `NodeEnv.capture_steps`: `replay()` issues one replay per device from this thread."""

    def __init__(self, node: NodeEnv, graphs, actions):
        self.node, self.graphs, self.actions = node, graphs, actions

    def replay(self):
        for sh, g in zip(self.node.shards, self.graphs):
            if sh.device.type == "cuda":
                with torch.cuda.device(sh.device):
                    g.replay()
            else:
                g.replay()
