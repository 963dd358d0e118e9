#!/usr/bin/env python3
"""C3 (BlockedUnlockPickup, 16384 envs) with every episode start generated on the device: us per step (hipGraph) by staging mode of
the truncation resets (BatchedMultiGridEnv.set_layout_generator(staged=...)), episodes in phase (the bench's start: all envs at step 0)
and out of phase (uniform over the episode length: the steady state of a long rollout), against the host-made layout pool."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from multigrid_amd import workloads
dev = torch.device("cuda:0")
wl = workloads.make("c3")
def run(staged, steady, lead=None):
    env = wl.make_env(dev, auto_reset=True)
    if staged == "pool":
        pass
    else:
        env.set_layout_generator("blockedunlockpickup", layout_seed=5, room_size=6, staged=staged, lead=lead)
    if steady:
        env.step_count.copy_(torch.arange(wl.batch, device=dev, dtype=torch.int32) % wl.spec.max_steps)
    m = bench.measure_steps(env, 256, 50, "graph", lambda: None, seed=4321, min_region_ms=30.0)
    return m["event_ms"] / m["timed_steps"] * 1e3, int(env.episode.sum())
for steady in (False, True):
    for staged, lead in (("pool", None), (False, None), (True, None), ("side", 16), ("side", 32), ("between", 8), ("between", 16), ("between", 32), ("between", 64)):
        us, ep = run(staged, steady, lead)
        print(f"{'out of phase' if steady else 'in phase    '}  staged={str(staged):8s} lead={lead}  {us:6.2f} us/step   episodes ended {ep}", flush=True)
