#!/usr/bin/env python3
"""Fused-step time on the other BASELINE.json configurations (C3 BlockedUnlockPickup, C5 64x64 / 16 agents / v=9) and
the large-batch bandwidth points for them (profiling aid; the bench line itself is C2).  GPU box."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from multigrid_amd import BatchedMultiGridEnv, EnvSpec, layouts  # noqa: E402

dev = torch.device("cuda", 0)


def c3(B):
    spec = EnvSpec(width=11, height=6, num_agents=2, view_size=7, max_steps=16 * 6 * 6, joint_reward=True,
                   env_kind="blockedunlockpickup")
    rng = np.random.default_rng(3)
    K = 256
    gs, ags, auxs = [], [], []
    for k in range(K):
        g, a, target = layouts.blockedunlockpickup_layout(6, 2, np.random.default_rng(1000 + k), np.random.default_rng(2000 + k))
        gs.append(g); ags.append(a); auxs.append(layouts.make_aux("blockedunlockpickup", g, target=target))
    idx = rng.integers(0, K, size=B)
    env = BatchedMultiGridEnv(spec, B, dev)
    env.load_state(np.stack(gs)[idx], np.stack(ags)[idx], aux=np.stack(auxs)[idx])
    env.seed_synthetic(7)
    return spec, env


def c5(B):
    spec = EnvSpec(width=64, height=64, num_agents=16, view_size=9, max_steps=4 * 64 * 64, env_kind="empty")
    r = np.random.default_rng(5)
    grid, agents = layouts.empty_layout(64, 16)
    agents = np.broadcast_to(agents, (B,) + agents.shape).copy()
    agents[..., 2] = r.integers(1, 63, size=(B, 16)); agents[..., 3] = r.integers(1, 63, size=(B, 16))
    agents[..., 1] = r.integers(0, 4, size=(B, 16))
    env = BatchedMultiGridEnv(spec, B, dev)
    env.load_state(np.broadcast_to(grid, (B,) + grid.shape).copy(), agents)
    env.seed_synthetic(7)
    return spec, env


for name, mk, batches in (("C3 BlockedUnlockPickup 11x6 A=2 v=7", c3, (16384, 1 << 20)),
                          ("C5 64x64 A=16 v=9", c5, (4096, 32768))):
    for B in batches:
        spec, env = mk(B)
        A = spec.num_agents
        acts = bench.random_actions(8, B, A, dev, 7)
        i = [0]

        def step():
            env.step(acts[i[0] & 7]); i[0] += 1
        t = bench.kernel_time_ms(step, 40, dev)
        o = bench.kernel_time_ms(env.gen_obs, 40, dev)
        env.check_errors()
        bs, bo = spec.bytes_step(), spec.bytes_gen_obs()
        print(f"{name} B={B}: step {t*1e3:.1f} us = {B*A/t*1e3:.3e} agent-steps/s, {B*A*bs/t/1e6:.0f} GB/s alg "
              f"({B*A*bs/t/1e6/8000:.3f} of peak) | gen_obs {o*1e3:.1f} us, {B*A*bo/o/1e6:.0f} GB/s ({B*A*bo/o/1e6/8000:.3f}) | "
              f"{env.backend.launch_info(B)}")
        del env
