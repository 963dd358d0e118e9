#!/usr/bin/env python3
"""Run W untimed + N profiled-looking fused steps (+ gen_obs) of the MGX_WORKLOAD configuration at a given batch; the target
of the rocprofv3 passes (tools/profile_round.py, tools/valu_by_phase.sh).
    python tools/large_step.py [batch] [N] [warm]
MGX_SKIP=<mask> (needs MGX_LIBMGX=multigrid_amd/lib/libmgx_dbg.so): skip phases -- results are then garbage."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
N = int(sys.argv[2]) if len(sys.argv) > 2 else 10
W = int(sys.argv[3]) if len(sys.argv) > 3 else 0
dev = torch.device("cuda", 0)
if os.environ.get("MGX_SKIP"):
    from multigrid_amd import _lib
    _lib.lib().mgx_debug_skip_phases(int(os.environ["MGX_SKIP"]))
spec = bench.workload_spec()
env = bench.make_env(spec, B, dev, 0)
acts = bench.random_actions(4, B, spec.num_agents, dev, 7)
if os.environ.get("MGX_GRAPH"):                 # the steps as hipGraph replays (what bench.py times): W + N launches in all
    for t in range(8):
        env.step(acts[t & 3], auto_reset=bench.AUTO_RESET)
    a2 = bench.random_actions(W + N, B, spec.num_agents, dev, 8)
    graph = bench.capture_steps(env, a2)
    graph.replay()
else:
    for t in range(W + N):
        env.step(acts[t & 3], auto_reset=bench.AUTO_RESET)
for t in range(W + N):
    env.gen_obs()
if os.environ.get("MGX_ONE_HOT_STEP"):          # the step with fused one-hot output, and the standalone one-hot kernel
    for t in range(W + N):
        env.step(acts[t & 3], auto_reset=bench.AUTO_RESET, one_hot=True)
    for t in range(W + N):
        env.one_hot_obs()
torch.cuda.synchronize()
if not os.environ.get("MGX_SKIP"):
    env.check_errors()
print("ok", bench.tool_workload(), B, N, W)
