#!/usr/bin/env python3
"""Fused kernel time vs envs per wavefront x wavefronts per workgroup at a large batch (profiling aid)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from multigrid_amd import _lib
dev = torch.device("cuda", 0)
spec = bench.workload_spec()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
env = bench.make_env(spec, B, dev, 0)
acts = bench.random_actions(4, B, spec.num_agents, dev, 7)
i = [0]
def step():
    env.step(acts[i[0] & 3]); i[0] += 1
for G in [int(x) for x in os.environ.get("MGX_GS", "4,5,6,7,8").split(",")]:
    for wpb in [int(x) for x in os.environ.get("MGX_WPBS", "1,2,4").split(",")]:
        _lib.lib().mgx_debug_set_envs_per_wavefront(G)
        _lib.lib().mgx_debug_set_waves_per_workgroup(wpb)
        t = bench.kernel_time_ms(step, 20, dev) * 1e3
        o = bench.kernel_time_ms(env.gen_obs, 20, dev) * 1e3
        li = env.backend.launch_info(B)
        print(f"G={G} wpb={wpb}: step {t:.1f} us gen_obs {o:.1f} us lds/wg {li['lds_bytes']} waves/CU {(160*1024 // li['lds_bytes']) * wpb}")
