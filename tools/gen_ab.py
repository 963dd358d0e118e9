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
QUICK = (("pool", None), (False, None), ("between", None), ("candidates", 128), ("candidates", 256), ("candidates", None), ("pool", None), ("candidates", None))
if os.environ.get("MGX_GEN_AB_QUICK") == "2":
    QUICK = (("pool", None), ("candidates", 512), ("candidates", 384), ("candidates", 256), ("candidates", 512))
for steady in (False, True):
    for staged, lead in QUICK if os.environ.get("MGX_GEN_AB_QUICK") else (("pool", None), (False, None), ("candidates", None), ("candidates", 32), ("candidates", 256), ("between", None), ("between", 32),
                         ("candidates", None), ("between", None), ("pool", None)):
        us, ep = run(staged, steady, lead)
        print(f"{'out of phase' if steady else 'in phase    '}  staged={str(staged):8s} lead={lead}  {us:6.2f} us/step   episodes ended {ep}", flush=True)
