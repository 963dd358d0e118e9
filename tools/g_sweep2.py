#!/usr/bin/env python3
"""Fused kernel time for explicit (envs per wavefront, wavefronts per workgroup) pairs (profiling aid).
usage: python tools/g_sweep2.py B G:wpb [G:wpb ...]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from multigrid_amd import _lib
dev = torch.device("cuda", 0)
spec = bench.workload_spec()
B = int(sys.argv[1])
env = bench.make_env(spec, B, dev, 0)
acts = bench.random_actions(4, B, spec.num_agents, dev, 7)
i = [0]
def step():
    env.step(acts[i[0] & 3]); i[0] += 1
for pair in sys.argv[2:]:
    G, wpb = (int(x) for x in pair.split(":"))
    _lib.lib().mgx_debug_set_envs_per_wavefront(G)
    _lib.lib().mgx_debug_set_waves_per_workgroup(wpb)
    t = bench.kernel_time_ms(step, 20, dev) * 1e3
    o = bench.kernel_time_ms(env.gen_obs, 20, dev) * 1e3
    li = env.backend.launch_info(B)
    print(f"G={G} wpb={wpb}: step {t:.1f} us gen_obs {o:.1f} us lds/wg {li['lds_bytes']} envs/wave {li['envs_per_wavefront']} waves/CU {(160*1024 // li['lds_bytes']) * wpb}")
