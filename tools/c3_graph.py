#!/usr/bin/env python3
"""C3 (BlockedUnlockPickup, 16384 envs) per-step time as a hipGraph for several envs-per-wavefront choices (profiling aid)."""
import os, sys, importlib.util
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from multigrid_amd import _lib
spec_ = importlib.util.spec_from_file_location("config_sweep_mod", os.path.join(os.path.dirname(__file__), "config_sweep.py"))
src = open(os.path.join(os.path.dirname(__file__), "config_sweep.py")).read().split("\nfor name, mk, batches in")[0]
ns = {"__file__": os.path.join(os.path.dirname(os.path.abspath(__file__)), "config_sweep.py")}
exec(compile(src, "config_sweep_head", "exec"), ns)
dev = torch.device("cuda", 0)
B, K = int(sys.argv[1]) if len(sys.argv) > 1 else 16384, 300
for G in (0, 4, 8, 16):
    _lib.lib().mgx_debug_set_envs_per_wavefront(G)
    best = 1e9
    for rep in range(2):
        spec, env = ns["c3"](B)
        acts = bench.random_actions(K, B, spec.num_agents, dev, 7)
        bench.AUTO_RESET = False
        for t in range(10):
            env.step(acts[t])
        _, ms = bench.timed_rollout(env, acts, "graph", lambda: None)
        best = min(best, ms * 1e3 / K)
        li = env.backend.launch_info(B)
        del env
    print(f"C3 B={B} G={G}: {best:.2f} us/step  {li}")
