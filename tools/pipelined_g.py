#!/usr/bin/env python3
"""C4 as two chains (the bench's `pipelined`) by envs per wavefront of the half-batch launches (debug-knobs build: mgx_debug_set_envs_per_wavefront).
Usage: MGX_LIBMGX=multigrid_amd/lib/libmgx_dbg.so python tools/pipelined_g.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from multigrid_amd import _lib, workloads
dev = torch.device("cuda", 0)
wl = workloads.make("c4")
for G in (0, 8, 16, 0, 8):
    _lib.lib().mgx_debug_set_envs_per_wavefront(G)
    env = wl.make_env(dev, auto_reset=True)
    m = bench.measure_steps(env, 250, 20, "graph", lambda: None, seed=5, min_region_ms=30.0, sub_shards=2)
    print(f"G={G}: two chains {m['event_ms'] / m['timed_steps'] * 1e3:.2f} us per step", flush=True)
    del env
_lib.lib().mgx_debug_set_envs_per_wavefront(0)
