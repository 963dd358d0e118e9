#!/usr/bin/env python3
"""Do the load bursts of a SIMD's wavefronts have to collide?  Per-step time of a hipGraph of K fused steps when every wavefront
first sleeps (its hardware wave slot on the SIMD) x S x 64 cycles (tools' build: bits 16..24 of mgx_debug_skip_phases).  Results
stay bit-identical (checked).  Needs MGX_LIBMGX=multigrid_amd/lib/libmgx_dbg.so.
Usage (GPU box): MGX_WORKLOAD=c4 python tools/stagger_probe.py 65536"""
import os
import sys
import zlib

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from multigrid_amd import _lib  # noqa: E402

dev = torch.device("cuda", 0)
spec = bench.workload_spec()
lib = _lib.lib()
K = 200
A = spec.num_agents
units = [int(x) for x in os.environ.get("MGX_STAGGER", "0,4,8,12,16,24,32,48").split(",")]
for B in [int(x) for x in sys.argv[1:]] or [65536]:
    sums = set()
    for by2 in (0, 1):
        for S in units:
            if by2 and S == 0:
                continue
            lib.mgx_debug_skip_phases((S << 16) | (by2 << 24))
            env = bench.make_env(spec, B, dev, 0)
            acts = bench.random_actions(K, B, A, dev, 7)
            graph = bench.capture_steps(env, acts)
            best = 1e9
            for rep in range(4):
                _, ms = bench.timed_region(env, graph.replay, 2, lambda: None)
                best = min(best, ms * 1e3 / (2 * K))
            torch.cuda.synchronize()
            crc = 0
            for t in (env.cells, env.agents, env.rng, env.step_count, env.obs, env.reward, env.terminated, env.episode):
                crc = zlib.crc32(t.cpu().numpy().tobytes(), crc)
            sums.add(crc)
            print(f"{bench.tool_workload()} B={B} stagger {S:3d} x 64 cycles per {'(slot & 1)' if by2 else '(slot & 3)'}"
                  f" (~{S * 64 / 2400:.2f} us per slot): {best:7.2f} us/step", flush=True)
            del graph, env
    assert len(sums) == 1, "results differ between settings"
lib.mgx_debug_skip_phases(0)
