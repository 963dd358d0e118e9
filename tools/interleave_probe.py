#!/usr/bin/env python3
"""Does stepping a batch as P independent sub-shards on P streams (one hipGraph with P parallel chains of step launches)
overlap one sub-shard's load phase with another's compute phase?  (profiling aid)
Usage (GPU box): MGX_WORKLOAD=c4 python tools/interleave_probe.py 65536"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda", 0)
spec = bench.workload_spec()
if os.environ.get("MGX_G"):                       # needs MGX_LIBMGX=multigrid_amd/lib/libmgx_dbg.so
    from multigrid_amd import _lib
    _lib.lib().mgx_debug_set_envs_per_wavefront(int(os.environ["MGX_G"]))
for B in [int(x) for x in sys.argv[1:]] or [65536]:
    plans = [[B // P + (1 if i < B % P else 0) for i in range(P)] for P in (1, 2, 3, 4)]
    for f in [float(x) for x in os.environ.get("MGX_SPLITS", "").split(",") if x]:      # uneven 2-way splits (first fraction)
        a = int(B * f) // 64 * 64
        plans.append([a, B - a])
    K = int(os.environ.get("MGX_K", "256"))
    for sizes in plans:
        P = len(sizes)
        firsts = [sum(sizes[:i]) for i in range(P)]
        envs = [bench.make_env(spec, sizes[i], dev, firsts[i]) for i in range(P)]
        acts = [bench.random_actions(K, sizes[i], spec.num_agents, dev, 7 + i) for i in range(P)]
        main = torch.cuda.current_stream(dev)
        cap = torch.cuda.Stream(dev)
        sides = [torch.cuda.Stream(dev) for _ in range(P - 1)]
        for e, a in zip(envs, acts):
            for t in range(20):
                e.step(a[t], auto_reset=bench.AUTO_RESET)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        cap.wait_stream(main)
        with torch.cuda.stream(cap):
            with torch.cuda.graph(graph, stream=cap):
                for s in sides:
                    s.wait_stream(cap)
                for i, (e, a) in enumerate(zip(envs, acts)):
                    st = cap if i == 0 else sides[i - 1]
                    with torch.cuda.stream(st):
                        for t in range(K):
                            e.step(a[t], auto_reset=bench.AUTO_RESET)
                for s in sides:
                    cap.wait_stream(s)
        main.wait_stream(cap)
        best = 1e9
        for rep in range(4):
            torch.cuda.synchronize()
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
            for _ in range(4):
                graph.replay()
            ev1.record()
            torch.cuda.synchronize()
            best = min(best, ev0.elapsed_time(ev1) * 1e3 / (4 * K))
        print(f"B={B} sub-shards={sizes} K={K}: {best:7.2f} us per step of the whole batch  ({B * spec.num_agents / best * 1e6:.3e} agent-steps/s)")
        del graph, envs, acts
