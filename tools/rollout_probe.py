#!/usr/bin/env python3
"""mgx_rollout_autoreset (ONE launch, the envs' state resident in LDS between the steps) over batch sizes of the C4 shape, beside the
lock-step graph of one-step launches at the same batch: what does the resident form cost per step, and at which occupancy?

    python tools/rollout_probe.py [c4|c2|c3] [batch ...]

The rollout's obs[T] goes to fresh memory every step (a write stream to HBM: T x B x A x v x v x 3 bytes), the lock-step graph rewrites
one buffer that stays in the Infinity Cache: both are reported as they are.
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from multigrid_amd import _lib, workloads  # noqa: E402

dev = torch.device("cuda", 0)


def rollout_us(wl, T, reps=3):
    env = wl.make_env(dev, auto_reset=True)
    spec, B = wl.spec, wl.batch
    A, v = spec.num_agents, spec.view_size
    acts = bench.random_actions(T, B, A, dev, 1234)
    warm = bench.random_actions(60, B, A, dev, 99)
    for t in range(60):                                     # (leave the cold start: every agent on one cell)
        env.step(warm[t], auto_reset=True)
    out = {"obs": torch.empty((T, B, A, v, v, 3), dtype=torch.uint8, device=dev),
           "dir": torch.empty((T, B, A), dtype=torch.uint8, device=dev),
           "reward": torch.empty((T, B, A), dtype=torch.float64, device=dev),
           "terminated": torch.empty((T, B, A), dtype=torch.uint8, device=dev),
           "truncated": torch.empty((T, B), dtype=torch.uint8, device=dev),
           "was_reset": torch.empty((T, B), dtype=torch.uint8, device=dev)}
    for x in out.values():
        x.zero_()
    env.rollout(acts, out, auto_reset=True)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        env.rollout(acts, out, auto_reset=True)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / T)
    env.check_errors()
    info = {}
    del env, out
    torch.cuda.empty_cache()
    return best, info


def graph_us(wl, T=200):
    env = wl.make_env(dev, auto_reset=True)
    B, A = wl.batch, wl.spec.num_agents
    acts = bench.random_actions(T, B, A, dev, 1234)
    for t in range(60):
        env.step(acts[t % T], auto_reset=True)
    g = env.capture_steps(acts, auto_reset=True)
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    del env
    return e0.elapsed_time(e1) * 1e3 / (3 * T)


def main():
    name = next((a for a in sys.argv[1:] if a in ("c2", "c3", "c4", "c5")), "c4")
    batches = [int(a) for a in sys.argv[1:] if a.isdigit()] or [4096, 8192, 16384, 24576, 32768, 49152, 65536]
    print(f"# library {_lib.LIB_PATH}; workload shape {name}; us per step of the batch")
    for B in batches:
        wl = workloads.make(name, batch=B, global_batch=max(B, workloads.GLOBAL_BATCH[name]))
        spec = wl.spec
        per_step = B * spec.num_agents * spec.view_size ** 2 * 3
        T = max(4, min(int(os.environ.get("MGX_T", "64")), (3 << 29) // per_step))
        r, info = rollout_us(wl, T)
        try:
            g = graph_us(wl)
        except Exception as e:                              # noqa: BLE001
            g = float("nan")
            print("graph:", e)
        print(f"B={B:6d} T={T:3d}  rollout {r:7.2f} us/step ({B * spec.num_agents / r / 1e3:6.2f} G agent-steps/s)   "
              f"lock-step graph {g:7.2f} us/step   {info}")


if __name__ == "__main__":
    main()
