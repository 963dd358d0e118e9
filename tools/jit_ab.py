#!/usr/bin/env python3
"""Generic instantiation vs the shape-specialised one compiled at run time (multigrid_amd/jit.py), same process, same box:
hipGraph-timed us per step of the plain step with auto-reset, env ids of the reference that BASELINE.json does not name.

    python tools/jit_ab.py [batch]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import multigrid_amd as mg  # noqa: E402
from multigrid_amd import jit, layouts  # noqa: E402
from multigrid_amd.batched import BatchedMultiGridEnv  # noqa: E402

dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
CASES = [("MultiGrid-Empty-8x8-v0", dict(agents=2)), ("MultiGrid-Empty-6x6-v0", dict(agents=4)), ("MultiGrid-Empty-16x16-v0", dict(agents=2)),
         ("MultiGrid-RedBlueDoors-8x8-v0", dict(agents=2)), ("MultiGrid-LockedHallway-4Rooms-v0", dict(agents=2)),
         ("MultiGrid-Playground-v0", dict(agents=4))]


def make(env_id, kw, specialise):
    spec = mg.spec_for(env_id, **kw)
    cls, cfg = mg.CONFIGURATIONS[env_id]
    host = cls(**{**cfg, **kw}, device="cuda:0", layout_seed=1)
    K = 64
    gs, ags, auxs = [], [], []
    for k in range(K):
        g, a, aux = host._gen_layout(host._layout_rng, np.random.default_rng(k))
        gs.append(g); ags.append(a); auxs.append(aux)
    env = BatchedMultiGridEnv(spec, B, dev, specialise=specialise)
    idx = np.arange(B) % K
    env.load_state(np.stack(gs)[idx], np.stack(ags)[idx], aux=None if auxs[0] is None else np.stack(auxs)[idx])
    env.seed_synthetic(3)
    env.set_layout_pool(np.stack(gs), np.stack(ags), None if auxs[0] is None else np.stack(auxs))
    return spec, env


def timed(env, spec, T=400):
    acts = bench.random_actions(T, B, spec.num_agents, dev, 5)
    for t in range(50):
        env.step(acts[t], auto_reset=True)
    g = env.capture_steps(acts, auto_reset=True)
    g.replay()
    return bench.kernel_time_ms(g.replay, 5, dev, warm=1) * 1e3 / T


print(f"# {B} envs, plain step with fused auto-reset (K = 64 layouts), hipGraph of 400 steps; us per step")
for env_id, kw in CASES:
    spec, env = make(env_id, kw, False)
    li = env.backend.launch_info(B)
    t_gen = timed(env, spec)
    t0 = time.perf_counter()
    spec, env2 = make(env_id, kw, True)
    t_c = time.perf_counter() - t0
    li2 = env2.backend.launch_info(B)
    t_jit = timed(env2, spec)
    print(f"{env_id:36s} A={spec.num_agents} {spec.width}x{spec.height}  envs/wave {li['envs_per_wavefront']:2d}  generic {t_gen:6.2f}  "
          f"specialised {t_jit:6.2f} ({100 * (t_jit / t_gen - 1):+5.1f} %)  [{env2.shape_kernel}, fixed_shape {li2['fixed_shape']}, setup {t_c:4.1f} s]")
