#!/usr/bin/env python3
"""Latency regime: per-step time of a hipGraph of K fused steps vs the small-group instantiation (slots per group 4 / 8 / 16,
one group per wavefront for 4 and 8) and, for 16, envs per wavefront.  Every setting must leave bit-identical state and
outputs (checked here: a wrong kernel would otherwise look fast).  Needs MGX_LIBMGX=multigrid_amd/lib/libmgx_dbg.so.
Usage (GPU box): MGX_WORKLOAD=c2 python tools/group_sweep.py 1024 2048 4096 8192 16384"""
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
K = 400
A = spec.num_agents
# (group, envs per wavefront): 0 = the library's own choice
settings = [(0, 0), (16, 0)] + [(g, 0) for g in (4, 8) if A <= g] + [(16, G) for G in (1, 2, 4, 8) if G * A <= 32]
for B in [int(x) for x in sys.argv[1:]] or [4096]:
    sums = {}
    for grp, G in settings:
        lib.mgx_debug_set_group(grp)
        lib.mgx_debug_set_envs_per_wavefront(G)
        env = bench.make_env(spec, B, dev, 0)
        acts = bench.random_actions(K, B, A, dev, 7)
        graph = bench.capture_steps(env, acts)
        best = 1e9
        for rep in range(4):
            _, ms = bench.timed_region(env, graph.replay, 2, lambda: None)
            best = min(best, ms * 1e3 / (2 * K))
        torch.cuda.synchronize()
        env.check_errors()
        crc = 0
        for t in (env.cells, env.agents, env.rng, env.step_count, env.obs, env.reward, env.terminated, env.episode):
            crc = zlib.crc32(t.cpu().numpy().tobytes(), crc)
        sums[(grp, G)] = crc
        li = env.backend.launch_info(B)
        print(f"{bench.tool_workload()} B={B} group={grp or 'auto'} G={G or 'auto'}: {best:7.2f} us/step   "
              f"[{li['envs_per_wavefront']} envs/wave, {li['slots_per_group']} slots/group, {li['workgroups']} x {li['threads_per_workgroup']}]"
              f"  crc {crc:08x}", flush=True)
        del graph, env
    assert len(set(sums.values())) == 1, f"B={B}: results differ between settings: {sums}"
lib.mgx_debug_set_group(0); lib.mgx_debug_set_envs_per_wavefront(0)
print("all settings bit-identical")
