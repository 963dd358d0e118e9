#!/usr/bin/env python3
"""The C4 step on the reference's byte grid u8[B,H,W,3] (MgxSpec.cell_bytes = 3: packed by the step kernel's own load phase) against
packed 16-bit cells: through torch.ops.mgx.step_out (eager, what bench.py reports as byte_grid_overhead) and as hipGraph replays of
BatchedMultiGridEnv.step (the kernels alone)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from multigrid_amd import workloads  # noqa: E402

dev = torch.device("cuda", 0)
for cb in (2, 3, 2, 3):
    wl = workloads.make("c4", cell_bytes=cb)
    env = wl.make_env(dev, auto_reset=True)
    m = bench.measure_steps(env, 64, 20, "graph", lambda: None, seed=7, min_region_ms=40.0)
    print(f"graph replays, cell_bytes={cb}: {m['event_ms'] / m['timed_steps'] * 1e3:.2f} us per step", flush=True)
    del env
print(json.dumps(bench.byte_grid_overhead(workloads.make("c4"), dev)))
