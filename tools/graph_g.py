#!/usr/bin/env python3
"""Per-step time of a hipGraph of K fused steps vs envs per wavefront / wavefronts per workgroup (profiling aid).
Usage (GPU box): python tools/graph_g.py [batch ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from multigrid_amd import _lib  # noqa: E402

dev = torch.device("cuda", 0)
spec = bench.workload_spec()
K = 500
for B in [int(x) for x in sys.argv[1:]] or [4096, 16384, 65536]:
    env = bench.make_env(spec, B, dev, 0)
    acts = bench.random_actions(K, B, spec.num_agents, dev, 7)
    for G in [int(x) for x in os.environ.get("MGX_GS", "0,1,2,3,4,8").split(",")]:
        for wpb in [int(x) for x in os.environ.get("MGX_WPBS", "1,2,4").split(",")]:
            _lib.lib().mgx_debug_set_envs_per_wavefront(G)
            _lib.lib().mgx_debug_set_waves_per_workgroup(wpb)
            best = 1e9
            graph = bench.capture_steps(env, acts)          # (the launch geometry is fixed at capture)
            for rep in range(3):
                _, ms = bench.timed_region(env, graph.replay, 2, lambda: None)
                best = min(best, ms * 1e3 / (2 * K))
            del graph
            li = env.backend.launch_info(B)
            print(f"B={B} G={G} wpb={wpb}: {best:7.2f} us/step  wg={li['workgroups']} x {li['threads_per_workgroup']}")
    _lib.lib().mgx_debug_set_envs_per_wavefront(0); _lib.lib().mgx_debug_set_waves_per_workgroup(0)
    del env
