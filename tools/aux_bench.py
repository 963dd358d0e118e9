#!/usr/bin/env python3
"""Bandwidth of the kernels either side of the fused step (one-hot, fully-observable, auto-reset) at HBM-resident sizes.
GPU box.  Prints one JSON object (also used by bench.py's `aux_kernels` leg)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


if __name__ == "__main__":
    print(json.dumps(bench.aux_kernel_points(torch.device("cuda", 0), int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20), indent=1))
