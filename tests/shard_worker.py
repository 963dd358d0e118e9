"""Worker of tests/test_multirank_hip.py: one rank of `python -m torch.distributed.run`, stepping its shard of a
BASELINE workload on the HIP backend (every rank on device 0 when MGX_BENCH_ONE_GPU=1: the single-GPU box has no
second device), and writing what it computed to an .npz for the parent to compare."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from multigrid_amd import workloads  # noqa: E402
from multigrid_amd.sharding import rank_world_from_env, shard_range  # noqa: E402


def actions_for(name, T, G, A):
    return np.stack([np.random.default_rng(900 + t).integers(0, 7, size=(G, A)).astype(np.int8) for t in range(T)])


def run_shard(name, G, T, first, count, device):
    wl = workloads.make(name, batch=count, first_env=first, global_batch=G)
    env = wl.make_env(device, auto_reset=True)
    acts = actions_for(name, T, G, wl.spec.num_agents)
    sums = []
    for t in range(T):
        o = env.step(torch.from_numpy(np.ascontiguousarray(acts[t, first:first + count])).to(device), auto_reset=True)
        sums.append([x.clone() for x in o])
    env.check_errors()
    last = {k: v.cpu().numpy() for k, v in zip(("obs", "dir", "reward", "terminated", "truncated"), sums[-1])}
    mid = sums[T // 2][0].cpu().numpy()
    return dict(last, obs_mid=mid, grid=env.grid.cpu().numpy(), agents=env.agents.cpu().numpy(),
                rng=env.rng.cpu().numpy(), step_count=env.step_count.cpu().numpy(), episode=env.episode.cpu().numpy())


def main():
    name, G, T, outdir = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    rank, local_rank, world = rank_world_from_env()
    one_gpu = os.environ.get("MGX_BENCH_ONE_GPU") == "1"
    device = torch.device("cuda", 0 if one_gpu else local_rank)
    torch.cuda.set_device(device)
    dist.init_process_group("gloo" if one_gpu else "nccl")
    first, count = shard_range(G, rank, world)
    res = run_shard(name, G, T, first, count, device)
    np.savez(os.path.join(outdir, f"shard{rank}.npz"), first=first, count=count, **res)
    t = torch.tensor([float(rank + 1)], dtype=torch.float64, device="cpu" if one_gpu else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)                 # what bench.py does with the elapsed time
    assert t.item() == world
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
