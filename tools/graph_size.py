#!/usr/bin/env python3
"""Does the per-step time of a hipGraph replay depend on the graph's size or on how far the simulation has run?
(profiling aid)  Replays graphs of K steps back to back on one env and prints the per-step time of every replay."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda", 0)
spec = bench.workload_spec()
B = 4096
for K, reps in ((100, 12), (200, 8), (1000, 3)):
    env = bench.make_env(spec, B, dev, 0)
    acts = bench.random_actions(K, B, spec.num_agents, dev, 7)
    for t in range(50):
        env.step(acts[t % K], auto_reset=bench.AUTO_RESET)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream(dev)
    s.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for t in range(K):
                env.step(acts[t], auto_reset=bench.AUTO_RESET)
    torch.cuda.current_stream(dev).wait_stream(s)
    out = []
    for r in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record(); g.replay(); e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) * 1e3 / K)
    print(f"K={K}: " + " ".join(f"{x:.2f}" for x in out))
    # the same replays without a sync in between
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for r in range(reps):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    print(f"K={K}: {reps} replays back to back: {e0.elapsed_time(e1) * 1e3 / (K * reps):.2f} us/step")
    del env, g
