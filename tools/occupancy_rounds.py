#!/usr/bin/env python3
"""C4 (or MGX_WORKLOAD) per-step time of a hipGraph of K fused steps with the wavefronts a CU holds at a time LIMITED by unused LDS
(debug-knobs build: mgx_debug_set_lds_pad) -- does a launch that runs as two rounds of half the wavefronts, the second round's loads
and step logic under the first round's stores, beat one round of all of them?
Usage (GPU box): MGX_LIBMGX=multigrid_amd/lib/libmgx_dbg.so MGX_WORKLOAD=c4 python tools/occupancy_rounds.py [batch]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from multigrid_amd import _lib  # noqa: E402

dev = torch.device("cuda", 0)
spec = bench.workload_spec()
K = 300
L = _lib.lib()
for B in [int(x) for x in sys.argv[1:]] or [65536]:
    env = bench.make_env(spec, B, dev, 0)
    acts = bench.random_actions(K, B, spec.num_agents, dev, 7)
    for wpb in (1, 4):
        L.mgx_debug_set_waves_per_workgroup(wpb)
        li = env.backend.launch_info(B)
        per_wg = li["lds_bytes"]
        for waves_per_cu in (16, 14, 12, 10, 8, 6, 4):
            wgs = waves_per_cu // wpb
            if wgs * wpb != waves_per_cu:
                continue
            pad = max(0, (160 * 1024) // wgs - per_wg - 256) if waves_per_cu < 16 else 0
            if per_wg + pad > 160 * 1024:
                continue
            L.mgx_debug_set_lds_pad(pad)
            best = 1e9
            try:
                graph = bench.capture_steps(env, acts)
                for rep in range(3):
                    _, ms = bench.timed_region(env, graph.replay, 2, lambda: None)
                    best = min(best, ms * 1e3 / (2 * K))
                del graph
            except Exception as e:      # noqa: BLE001
                print(f"B={B} wpb={wpb} waves/CU<={waves_per_cu}: {type(e).__name__} {str(e)[:80]}")
                continue
            print(f"B={B} wpb={wpb} wavefronts per CU <= {waves_per_cu:2d} (LDS {per_wg}+{pad} B per workgroup): {best:7.2f} us/step", flush=True)
        L.mgx_debug_set_lds_pad(0)
    L.mgx_debug_set_waves_per_workgroup(0)
    del env
