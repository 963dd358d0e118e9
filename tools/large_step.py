#!/usr/bin/env python3
"""Run N fused steps (+N gen_obs) of the bench workload at a given per-GPU batch; target for rocprofv3 passes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
N = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda", 0)
if os.environ.get("MGX_SKIP"):                      # profiling: skip phases (results are then garbage)
    from multigrid_amd import _lib
    _lib.lib().mgx_debug_skip_phases(int(os.environ["MGX_SKIP"]))
spec = bench.workload_spec()
env = bench.make_env(spec, B, dev, 0)
acts = bench.random_actions(4, B, spec.num_agents, dev, 7)
for t in range(N):
    env.step(acts[t & 3], auto_reset=bench.AUTO_RESET)
for t in range(N):
    env.gen_obs()
torch.cuda.synchronize()
if not os.environ.get("MGX_SKIP"):
    env.check_errors()
print("ok", B, N)
