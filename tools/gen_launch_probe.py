#!/usr/bin/env python3
"""How long one generator launch (mgx_stage_generate, MgxGenStage.candidates) takes at C3 (16384 BlockedUnlockPickup envs), by how
many envs restarted since the previous one.  Usage: python tools/gen_launch_probe.py"""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multigrid_amd import _lib, workloads
from multigrid_amd.ops import _stream

dev = torch.device("cuda:0")
wl = workloads.make("c3")
g = torch.Generator(device=dev); g.manual_seed(1)
for steady, gap in ((True, 16), (True, 64), (True, 128), (False, 64), (True, 0)):
    env = wl.make_env(dev, auto_reset=True)
    env.set_layout_generator("blockedunlockpickup", layout_seed=5, room_size=6, staged="candidates", lead=2 * max(gap, 1))
    if steady:
        env.step_count.copy_(torch.arange(wl.batch, device=dev, dtype=torch.int32) % wl.spec.max_steps)
    st = env._gen["stage"]
    gen_c = env.backend._layout_gen_struct(env._gen)
    times, todo = [], []
    for rep in range(6):
        st["lead"] = 500; st["phase"][0] = 1                   # (no generator launch from step() within `gap` steps)
        env._bound.clear()
        for t in range(gap):
            env.step(torch.randint(0, 7, (wl.batch, wl.spec.num_agents), dtype=torch.int8, device=dev, generator=g), auto_reset=True)
        todo.append(int((st["tag"] != env.episode[:, None]).sum()))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        rc = _lib.lib().mgx_stage_generate(C.byref(env.backend.sc), wl.batch, C.byref(gen_c), env.rng.data_ptr(), env.episode.data_ptr(), _stream(dev))
        e1.record()
        torch.cuda.synchronize()
        assert rc == 0, rc
        times.append(e0.elapsed_time(e1) * 1e3)
    print(f"{'out of phase' if steady else 'in phase    '} {gap:4d} steps between launches: candidates to make {todo}  launch us {[round(x, 1) for x in times]}", flush=True)
