#!/usr/bin/env python3
"""Same-box A/B of two builds of the library on the bench's points (kernel time per step as hipGraph replays):
    python tools/ab_point.py libA.so libB.so [libC.so ...] [rounds]     -> alternating runs, C4 / C2 / C3 lock step + the resident rollout at 49152 envs
Each measurement runs in its own process (the library is chosen at import: MGX_LIBMGX)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import os, sys, torch
sys.path.insert(0, %r)
import bench
from multigrid_amd import workloads
dev = torch.device("cuda", 0)
out = []
for name, B in (("c4", 65536), ("c2", 4096), ("c3", 16384), ("c4", 8192)) + ((("c5", 32768),) if os.environ.get("AB_C5") else ()):
    wl = workloads.make(name, batch=B, global_batch=max(B, workloads.GLOBAL_BATCH[name]))
    env = wl.make_env(dev, auto_reset=True)
    T = 256
    acts = bench.random_actions(T, B, wl.spec.num_agents, dev, 1234)
    for t in range(60): env.step(acts[t], auto_reset=True)
    g = env.capture_steps(acts, auto_reset=True)
    g.replay()
    ms = bench.kernel_time_ms(g.replay, 8, dev, warm=2) / T
    out.append(f"{name}@{B} {ms * 1e3:.3f}")
    del env, g
wl = workloads.make("c4", batch=49152, global_batch=65536)
env = wl.make_env(dev, auto_reset=True)
acts = bench.random_actions(48, 49152, 4, dev, 1)
for t in range(48): env.step(acts[t], auto_reset=True)
o = env.rollout(acts, auto_reset=True)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
best = 1e9
for r in range(3):
    e0.record(); env.rollout(acts, o, auto_reset=True); e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) * 1e3 / 48)
out.append(f"rollout49152 {best:.3f}")
print(" | ".join(out))
''' % ROOT
libs = [a for a in sys.argv[1:] if a.endswith(".so")]          # two or more builds, measured in alternation
rounds = next((int(a) for a in sys.argv[1:] if a.isdigit()), 3)
for r in range(rounds):
    for lib in libs:
        p = subprocess.run([sys.executable, "-c", CODE], env=dict(os.environ, MGX_LIBMGX=os.path.join(ROOT, lib)), capture_output=True, text=True)
        print(f"{os.path.basename(lib):16s} {p.stdout.strip().splitlines()[-1] if p.stdout.strip() else p.stderr[-300:]}", flush=True)
