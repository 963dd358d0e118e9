#!/usr/bin/env python3
"""us per step of the batch for P = 1, 2, 3, 4, 6, 8 chains (capture_steps(sub_shards=P), hipGraph replays, product library):
what mgx_sub_shards' answer should be for a shape.   MGX_WORKLOAD=c4 python tools/chain_sweep.py [batch]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda", 0)
spec = bench.workload_spec()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
for rep in range(2):
    for P in (1, 2, 3, 4, 6, 8):
        env = bench.make_env(spec, B, dev, 0)
        m = bench.measure_steps(env, 64, 20, "graph", lambda: None, seed=7, min_region_ms=40.0, sub_shards=P)
        print(f"{bench.tool_workload()} B={B} P={P}: {m['event_ms'] / m['timed_steps'] * 1e3:.2f} us per step (hint {env.sub_shards_hint(True)})", flush=True)
        del env
