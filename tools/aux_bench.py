#!/usr/bin/env python3
"""Bandwidth of the kernels either side of the fused step (one-hot, fully-observable, auto-reset) at HBM-resident sizes.
GPU box.  Prints one JSON object (bench.py's `aux_kernels` leg).  Usage: python tools/aux_bench.py [envs]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from multigrid_amd import workloads  # noqa: E402


if __name__ == "__main__":
    dev = torch.device("cuda", 0)
    wl = workloads.make("c4")
    env = bench.make_env(wl.spec, int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20, dev, 0)
    env.step(bench.random_actions(1, env.batch, wl.spec.num_agents, dev, 3)[0], auto_reset=bench.AUTO_RESET)
    print(json.dumps(bench.aux_kernel_points(env, dev), indent=1))
