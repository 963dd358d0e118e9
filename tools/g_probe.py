#!/usr/bin/env python3
"""Kernel time vs envs-per-workgroup (profiling aid)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from multigrid_amd import _lib
dev = torch.device("cuda", 0)
spec = bench.workload_spec()
for B in [int(x) for x in sys.argv[1:]] or [1 << 20]:
    env = bench.make_env(spec, B, dev, 0)
    acts = bench.random_actions(4, B, spec.num_agents, dev, 7)
    i = [0]
    def step():
        env.step(acts[i[0] & 3]); i[0] += 1
    for G in (4, 8, 16):
        _lib.lib().mgx_debug_set_envs_per_wavefront(G)
        t = bench.kernel_time_ms(step, 30, dev) * 1e3
        o = bench.kernel_time_ms(env.gen_obs, 30, dev) * 1e3
        print(f"B={B} G={G}: step {t:.1f} us  gen_obs {o:.1f} us  {env.backend.launch_info(B)}")
    _lib.lib().mgx_debug_set_envs_per_wavefront(0)
for B in [1 << 20]:
    env = bench.make_env(spec, B, dev, 0)
    acts = bench.random_actions(4, B, spec.num_agents, dev, 7)
    i = [0]
    def step():
        env.step(acts[i[0] & 3]); i[0] += 1
    _lib.lib().mgx_debug_set_envs_per_wavefront(8)
    for wpb in (1, 2, 3, 4):
        _lib.lib().mgx_debug_set_waves_per_workgroup(wpb)
        t = bench.kernel_time_ms(step, 20, dev) * 1e3
        o = bench.kernel_time_ms(env.gen_obs, 20, dev) * 1e3
        print(f"B={B} wpb={wpb}: step {t:.1f} us  gen_obs {o:.1f} us  {env.backend.launch_info(B)}")
    _lib.lib().mgx_debug_set_envs_per_wavefront(0); _lib.lib().mgx_debug_set_waves_per_workgroup(0)
