#!/usr/bin/env python3
"""Per-step time of a hipGraph of K fused steps for explicit (envs per wavefront, wavefronts per workgroup) pairs, each on
a FRESH env (profiling aid).  usage: python tools/graph_g2.py B G:wpb [G:wpb ...]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from multigrid_amd import _lib
dev = torch.device("cuda", 0)
spec = bench.workload_spec()
B = int(sys.argv[1]); K = 400
acts = bench.random_actions(K, B, spec.num_agents, dev, 7)
for pair in sys.argv[2:]:
    G, wpb = (int(x) for x in pair.split(":"))
    _lib.lib().mgx_debug_set_envs_per_wavefront(G)
    _lib.lib().mgx_debug_set_waves_per_workgroup(wpb)
    best = 1e9
    for rep in range(3):
        env = bench.make_env(spec, B, dev, 0)
        for t in range(20):
            env.step(acts[t], auto_reset=bench.AUTO_RESET)
        _, ms = bench.timed_rollout(env, acts, "graph", lambda: None)
        best = min(best, ms * 1e3 / K)
        li = env.backend.launch_info(B)
        del env
    print(f"B={B} G={G} wpb={wpb}: {best:7.2f} us/step  wg={li['workgroups']} x {li['threads_per_workgroup']}")
