#!/usr/bin/env python3
"""Same-box A/B of mgx_full_obs between two builds of the library (MGX_LIBMGX chosen at import: one process per measurement):
    python tools/full_obs_ab.py libA.so libB.so [rounds]
Points: 1 M envs of the C4 shape (16x16, 16-bit cells: the bench's `full_obs` row), 32768 envs of C5's (64x64, compact and 16-bit
cells), 262144 envs of an odd shape (9x7, 3 agents: slices that start off a dword take the input-ordered pass)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, torch
sys.path.insert(0, %r)
import bench
from multigrid_amd import workloads
from multigrid_amd.spec import EnvSpec
from multigrid_amd.batched import BatchedMultiGridEnv
dev = torch.device("cuda", 0)
out = []
def point(tag, env):
    B, sp = env.batch, env.spec
    cb = 1 if sp.cell_bytes == 1 else 2
    nbytes = B * (sp.height * sp.width * (cb + 3) + sp.num_agents * 8)
    ms = min(bench.kernel_time_ms(env.full_obs, 30, dev, warm=10) for _ in range(3))
    out.append(f"{tag} {ms * 1e3:.1f} us ({nbytes / ms / 1e6 / 8e3:.3f})")
wl = workloads.make("c4", batch=1 << 20, global_batch=1 << 20)
point("c4x1M", wl.make_env(dev, auto_reset=False))
for cbytes in (1, 2):
    wl = workloads.make("c5", cell_bytes=cbytes)
    point(f"c5/cb{cbytes}", wl.make_env(dev, auto_reset=False))
import numpy as np
from multigrid_amd import layouts
sp = EnvSpec(9, 7, 3, 5, max_steps=20)
env = BatchedMultiGridEnv(sp, 262144, dev)
g = np.zeros((7, 9, 3), np.uint8); g[..., 0] = 1
g[0] = g[-1] = (2, 5, 0); g[:, 0] = g[:, -1] = (2, 5, 0)
ag = layouts._fresh_agents(3); ag[:, 2] = 0; ag[:, 3] = [1, 2, 3]; ag[:, 4] = 1
env.load_state(np.broadcast_to(g, (262144,) + g.shape).copy(), np.broadcast_to(layouts.pack_agents(ag), (262144, 3, 8)).copy(), validate=False)
point("9x7x3", env)
print(" | ".join(out))
''' % ROOT
libs = sys.argv[1:3]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 2
for r in range(rounds):
    for lib in libs:
        env = dict(os.environ, MGX_LIBMGX=os.path.abspath(lib))
        p = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True)
        print(f"{os.path.basename(lib):16s} {p.stdout.strip() or p.stderr.strip()[-400:]}", flush=True)
