#!/usr/bin/env python3
"""Attribute fused-kernel time to phases by skipping them (profiling aid; results of skipped runs are garbage).
Usage (GPU box): python tools/phase_probe.py [batch ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from multigrid_amd import _lib  # noqa: E402

dev = torch.device("cuda", 0)
spec = bench.workload_spec()
batches = [int(x) for x in sys.argv[1:]] or [4096, 65536, 1 << 20]
names = ["P0 load", "P1 step", "P2 gather", "P3 vis", "P4 mask", "P5 store"]
for B in batches:
    env = bench.make_env(spec, B, dev, 0)
    acts = bench.random_actions(4, B, spec.num_agents, dev, 7)
    i = [0]

    def step():
        env.step(acts[i[0] & 3]); i[0] += 1
    for t in range(20):
        step()
    _lib.lib().mgx_debug_skip_phases(0)
    full = bench.kernel_time_ms(step, 50, dev) * 1e3
    obs = bench.kernel_time_ms(env.gen_obs, 50, dev) * 1e3
    print(f"B={B}: full step {full:.2f} us, gen_obs {obs:.2f} us, launch {env.backend.launch_info(B)}")
    for p, n in enumerate(names):
        _lib.lib().mgx_debug_skip_phases(1 << p)
        t = bench.kernel_time_ms(step, 50, dev) * 1e3
        print(f"   skip {n:10s}: {t:8.2f} us  (delta {full - t:7.2f})")
    _lib.lib().mgx_debug_skip_phases(63)
    t = bench.kernel_time_ms(step, 50, dev) * 1e3
    print(f"   skip all       : {t:8.2f} us")
    _lib.lib().mgx_debug_skip_phases(0)
    del env
