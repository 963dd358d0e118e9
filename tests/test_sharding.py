"""N>1 path on CPU: two gloo ranks each step their shard of the batch; the concatenated results must be bit-identical
to one process stepping the whole batch (no collective is needed for the data path; gloo is only used to gather the
results for comparison, exactly as bench.py only uses RCCL for barrier/timing)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from multigrid_amd import BatchedMultiGridEnv, EnvSpec, layouts
from multigrid_amd.sharding import make_sharded_env, shard_range
from tests import util

SPEC = EnvSpec(16, 16, 4, 7, max_steps=1024)
GLOBAL_B, T, SEED = 37, 6, 11


def test_shard_range_partitions():
    for B in (0, 1, 7, 64, 65537):
        for R in (1, 2, 3, 8):
            blocks = [shard_range(B, r, R) for r in range(R)]
            assert sum(c for _, c in blocks) == B
            pos = 0
            for first, count in blocks:
                assert first == pos
                pos += count
            assert max(c for _, c in blocks) - min(c for _, c in blocks) <= 1
    with pytest.raises(ValueError):
        shard_range(8, 2, 2)


def _actions():
    return np.stack([util.random_actions(GLOBAL_B, SPEC.num_agents, seed=100 + t) for t in range(T)])


def _run(env, first, count):
    grid, agents = layouts.empty_layout(SPEC.width, SPEC.num_agents)
    env.load_state(grid, agents)
    env.seed_synthetic(SEED)                      # a function of the GLOBAL env index
    acts = _actions()
    outs = []
    for t in range(T):
        o = env.step(torch.from_numpy(np.ascontiguousarray(acts[t, first:first + count])))
        outs.append([x.clone() for x in o])
    return outs, env.grid.clone(), env.rng.clone()


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        first, count = shard_range(GLOBAL_B, rank, world)
        env = make_sharded_env(SPEC, GLOBAL_B, rank, world, device="cpu", backend=util.OracleBackend(SPEC))
        assert (env.first_env, env.batch) == (first, count)
        outs, grid, rng = _run(env, first, count)
        # gather every shard's final obs / rng on rank 0 (ragged: pad to the largest shard)
        mx = max(shard_range(GLOBAL_B, r, world)[1] for r in range(world))
        def gather(x):
            pad = torch.zeros((mx,) + tuple(x.shape[1:]), dtype=x.dtype)
            pad[:x.shape[0]] = x
            bufs = [torch.zeros_like(pad) for _ in range(world)]
            dist.all_gather(bufs, pad)
            return torch.cat([b[:shard_range(GLOBAL_B, r, world)[1]] for r, b in enumerate(bufs)])
        full = [[gather(x) for x in step] for step in outs]
        g, r = gather(grid), gather(rng)
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)           # what bench.py does with the elapsed time
        assert t.item() == world
        if rank == 0:
            q.put(([[x.numpy() for x in step] for step in full], g.numpy(), r.numpy()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_rank_shards_equal_single_process():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    sharded_outs, sharded_grid, sharded_rng = q.get()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    env = BatchedMultiGridEnv(SPEC, GLOBAL_B, "cpu", backend=util.OracleBackend(SPEC))
    outs, grid, rng = _run(env, 0, GLOBAL_B)
    for t in range(T):
        for a, b in zip(sharded_outs[t], outs[t]):
            assert a.tobytes() == b.numpy().tobytes()
    assert sharded_grid.tobytes() == grid.numpy().tobytes()
    assert sharded_rng.tobytes() == rng.numpy().tobytes()


def test_split_views_step_the_parent():
    """BatchedMultiGridEnv.split: sub-shards are views (no copies) whose steps, in any interleaving, are the parent's steps
    -- incl. the auto-reset layout choice, a function of the global env index.  (Oracle backend: host logic only.)"""
    spec = EnvSpec(8, 8, 2, 7, max_steps=4)
    B, K = 200, 3
    pool = util.random_state(spec, K, seed=3, terminated_p=0.0)
    st = util.random_state(spec, B, seed=4)

    def mk():
        e = BatchedMultiGridEnv(spec, B, "cpu", first_env=1000, backend=util.OracleBackend(spec))
        e.load_state(st["grid"], st["agents"], st["rng"], None, st["step_count"] % 4)
        e.set_layout_pool(pool["grid"], pool["agents"])
        return e
    whole, parts = mk(), mk()
    shards = parts.split(3)
    assert [s.batch for s in shards] == [128, 64, 8] and [s.first_env for s in shards] == [1000, 1128, 1192]
    assert shards[1].cells.data_ptr() == parts.cells[128:].data_ptr()                   # views, not copies
    for t in range(6):
        act = torch.from_numpy(util.random_actions(B, 2, seed=t, p_missing=0.0))
        whole.reset_done(); want = [x.clone() for x in whole.step(act)]
        for s in reversed(shards):                                                       # any order
            lo, hi = s._range
            s.reset_done(); s.step(act[lo:hi].contiguous())
        for x, y in zip((parts.obs, parts.dir, parts.reward, parts.terminated, parts.truncated), want):
            assert torch.equal(x, y)
        assert torch.equal(parts.cells, whole.cells) and torch.equal(parts.episode, whole.episode)
    assert int(whole.episode.sum()) > B


def _node_vs_whole(devices, backend_factory, whole_device, whole_backend):
    """NodeEnv over `devices` (eager steps, then a captured block) == one BatchedMultiGridEnv over the whole batch."""
    from multigrid_amd.sharding import NodeEnv
    spec = EnvSpec(8, 8, 2, 7, max_steps=5)
    B, K, T = 403, 3, 7
    pool = util.random_state(spec, K, seed=3, terminated_p=0.0)
    st = util.random_state(spec, B, seed=4)
    node = NodeEnv(spec, B, devices, backend_factory=backend_factory)
    assert [hi - lo for lo, hi in node.ranges] == [shard_range(B, r, len(devices))[1] for r in range(len(devices))]
    whole = BatchedMultiGridEnv(spec, B, whole_device, backend=whole_backend)
    for e in (node, whole):
        e.load_state(st["grid"], st["agents"], st["rng"], None, st["step_count"] % 5)
        e.set_layout_pool(pool["grid"], pool["agents"])
    acts = torch.from_numpy(np.stack([util.random_actions(B, 2, seed=t, p_missing=0.0) for t in range(2 * T)]))
    for t in range(T):                                          # eager, one tensor over the global batch
        node.step(acts[t], auto_reset=True)
        whole.step(acts[t].to(whole.device), auto_reset=True)
    g = node.capture_steps(acts[T:], auto_reset=True)          # then T more as one graph per shard
    g.replay()
    for t in range(T, 2 * T):
        whole.step(acts[t].to(whole.device), auto_reset=True)
    for name in ("obs", "dir", "reward", "terminated", "truncated", "grid", "agents", "rng", "step_count", "episode"):
        assert torch.equal(node.gather(name), getattr(whole, name).cpu()), name
    node.check_errors()


def test_node_env_in_one_process_equals_one_env_on_cpu():
    """sharding.NodeEnv (round 6): the shards of a 3-way split held by ONE process, on the oracle backend."""
    _node_vs_whole(["cpu"] * 3, lambda spec, dev: util.OracleBackend(spec), "cpu", util.OracleBackend(EnvSpec(8, 8, 2, 7, max_steps=5)))


@pytest.mark.gpu
def test_node_env_in_one_process_equals_one_env_on_gpu():
    """... on the HIP kernels: eight shards (here all on the one device the box has; on a node: cuda:0 .. cuda:7), eager steps and
    one captured graph per shard replayed from a single thread."""
    n = torch.cuda.device_count()
    _node_vs_whole([f"cuda:{r % n}" for r in range(8)], None, "cuda:0", None)
