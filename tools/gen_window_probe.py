#!/usr/bin/env python3
"""C3 with every episode start generated on the device: us per step by WINDOW of 64 steps of the episode cycle (a 64-step hipGraph
replayed back to back, one event pair per replay) -- where in the cycle the time over the layout pool's goes.
Usage: python tools/gen_window_probe.py [staged [lead]]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from multigrid_amd import workloads
dev = torch.device("cuda:0")
wl = workloads.make("c3")
staged = sys.argv[1] if len(sys.argv) > 1 else "candidates"
lead = int(sys.argv[2]) if len(sys.argv) > 2 else None
K = 64
for steady in (False, True):
    env = wl.make_env(dev, auto_reset=True)
    if staged != "pool":
        env.set_layout_generator("blockedunlockpickup", layout_seed=5, room_size=6, staged=False if staged == "unstaged" else staged, lead=lead)
    if steady:
        env.step_count.copy_(torch.arange(wl.batch, device=dev, dtype=torch.int32) % wl.spec.max_steps)
    acts = bench.random_actions(K, wl.batch, wl.spec.num_agents, dev, 7)
    graph = env.capture_steps(acts, auto_reset=True)
    for _ in range(9):
        graph.replay()                                            # one whole cycle of warm-up
    n = 27
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    torch.cuda.synchronize()
    ev[0].record()
    for i in range(n):
        graph.replay()
        ev[i + 1].record()
    torch.cuda.synchronize()
    us = [ev[i].elapsed_time(ev[i + 1]) * 1e3 / K for i in range(n)]
    print(f"{staged} lead={lead} {'out of phase' if steady else 'in phase'}: us/step per 64-step window: " + " ".join(f"{u:.2f}" for u in us)
          + f"   mean {sum(us) / n:.2f}", flush=True)
