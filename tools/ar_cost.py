#!/usr/bin/env python3
"""Kernel time of the step with and without the fused auto-reset (profiling aid).  usage: python tools/ar_cost.py [batch ...]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda", 0)
spec = bench.workload_spec()
for B in [int(x) for x in sys.argv[1:]] or [65536, 1 << 20]:
    env = bench.make_env(spec, B, dev, 0)
    acts = bench.random_actions(4, B, spec.num_agents, dev, 7)
    i = [0]
    res = []
    for ar in (False, True, False, True):
        def step():
            env.step(acts[i[0] & 3], auto_reset=ar); i[0] += 1
        res.append(f"{'AR' if ar else 'no'} {bench.kernel_time_ms(step, 40, dev) * 1e3:.1f}")
    print(f"B={B}: " + " | ".join(res) + " us")
    del env
