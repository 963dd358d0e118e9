#!/usr/bin/env python3
"""Per-step time of a hipGraph of K fused steps, with phases skipped one at a time (profiling aid: results of skipped
runs are garbage).  This excludes the per-launch host overhead, so it shows the kernel's own
critical path at small batches.  Usage (GPU box): python tools/graph_phase.py [batch ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from multigrid_amd import _lib  # noqa: E402

dev = torch.device("cuda", 0)
spec = bench.workload_spec()
K = 200
names = [(0, "full"), (1, "P0 load"), (2, "P1 step"), (4, "P2 gather"), (8, "P3 vis"), (16, "P4 mask"), (32, "P5 store"),
         (64, "P1s eval"), (128, "P1a draw"), (63, "all")]
for B in [int(x) for x in sys.argv[1:]] or [4096]:
    env = bench.make_env(spec, B, dev, 0)
    acts = bench.random_actions(K, B, spec.num_agents, dev, 7)
    res = []
    for mask, name in names:
        _lib.lib().mgx_debug_skip_phases(mask)
        best = 1e9
        for rep in range(3):
            graph = bench.capture_steps(env, acts)
            _, ms = bench.timed_region(env, graph.replay, 1, lambda: None)
            best = min(best, ms * 1e3 / K)
        res.append((name, best))
    _lib.lib().mgx_debug_skip_phases(0)
    full = res[0][1]
    print(f"B={B} launch {env.backend.launch_info(B)}")
    for name, t in res:
        print(f"   skip {name:10s}: {t:7.2f} us/step (delta {full - t:6.2f})")
    del env
