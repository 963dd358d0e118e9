#!/usr/bin/env python3
"""Randomised soak of device-side episode generation (test infrastructure; GPU box): every staging protocol of
set_layout_generator -- candidates / between / in_launch, random cadences, eager steps, sub-shard chains, captured blocks -- against
the UNSTAGED run (every finished env generated in the tail of its step; that form is pinned to the oracle and the reference's reset
fixtures by tests/test_layout_gen.py) on random shapes, batches, episode lengths and injected early episode ends.
usage: python tools/fuzz_generate.py [seconds] [seed]     exits 1 on the first mismatch."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multigrid_amd import BatchedMultiGridEnv, EnvSpec, layouts  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
r = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = "cuda:0"
t_end = time.time() + budget
n_case = n_steps = n_early = 0
FIELDS = ("cells", "agents", "rng", "step_count", "aux", "episode")
while time.time() < t_end:
    kind = str(r.choice(["bup", "bup", "bup", "empty_random", "redbluedoors", "lockedhallway", "playground"]))
    A = int(r.choice([2, 2, 3, 4]))
    M = int(r.integers(3, 40))
    # (only layouts with room to spare: the rejection sampling of an over-full room never ends, on the device as in the reference)
    if kind == "bup":
        rs = int(r.choice([5, 6, 6, 7]))
        A = min(A, rs - 3)
        spec = EnvSpec(2 * rs - 1, rs, A, int(r.choice([5, 7])), max_steps=M, joint_reward=True, env_kind="blockedunlockpickup")
        gen = dict(kind="blockedunlockpickup", room_size=rs)
        g0, a0, t0 = layouts.blockedunlockpickup_layout(rs, A, np.random.default_rng(1), np.random.default_rng(2))
        state = dict(grid=g0, agents=a0, aux=layouts.make_aux("blockedunlockpickup", g0, target=t0))
    elif kind == "empty_random":
        n = int(r.integers(5, 12))
        spec = EnvSpec(n, n, A, 7, max_steps=M)
        gen = dict(kind="empty_random")
        g0, a0 = layouts.empty_layout(n, A)
        state = dict(grid=g0, agents=a0)
    elif kind == "redbluedoors":
        n = int(r.integers(6, 10))
        spec = EnvSpec(2 * n, n, A, 7, max_steps=M, joint_reward=True, failure_termination_mode="any", env_kind="redbluedoors")
        gen = dict(kind="redbluedoors")
        g0, a0 = layouts.redbluedoors_layout(n, A, np.random.default_rng(1))
        state = dict(grid=g0, agents=a0, aux=layouts.make_aux("redbluedoors", g0))
    elif kind == "lockedhallway":
        rs, rows = 5, int(r.choice([1, 2]))
        A = min(A, 3)
        spec = EnvSpec(3 * (rs - 1) + 1, rows * (rs - 1) + 1, A, 7, max_steps=M, joint_reward=True, env_kind="lockedhallway")
        gen = dict(kind="lockedhallway", room_size=rs, max_hallway_keys=1, max_keys_per_room=2)
        g0, a0 = layouts.lockedhallway_layout(2 * rows, rs, 1, 2, A, np.random.default_rng(1), np.random.default_rng(2))
        state = dict(grid=g0, agents=a0, aux=layouts.make_aux("lockedhallway", g0))
    else:
        rs = 7                                                   # (12 objects into 4 rooms: smaller rooms can overflow -- no layout exists)
        spec = EnvSpec(2 * (rs - 1) + 1, 2 * (rs - 1) + 1, A, 7, max_steps=M)
        gen = dict(kind="playground", room_size=rs)
        g0, a0 = layouts.playground_layout(rs, 2, 2, A, np.random.default_rng(1), np.random.default_rng(2))
        state = dict(grid=g0, agents=a0)
    B = int(r.choice([1, 7, 64, 65, 300, 1000, 2500]))
    staged = str(r.choice(["candidates", "candidates", "between", "in_launch"]))
    if staged == "candidates" and (kind == "playground" or (kind == "bup" and gen["room_size"] > 6)):
        staged = "between"
    lead = int(r.integers(2, max(3, M)))
    mode = str(r.choice(["eager", "eager", "chains", "graph"]))
    seed = int(r.integers(0, 1 << 30))

    def make(st):
        env = BatchedMultiGridEnv(spec, B, dev, first_env=int(seed % 1000))
        env.load_state(state["grid"], state["agents"], aux=state.get("aux"))
        env.seed_synthetic(seed % 97)
        env.set_layout_generator(layout_seed=seed % 89, staged=st, lead=lead, **gen)
        env.step_count.copy_(torch.from_numpy(np.random.default_rng(seed).integers(0, M, size=B).astype(np.int32)).to(dev))
        return env

    ref, env = make(False), make(staged)
    T = int(r.integers(M, 3 * M + 5))
    g = torch.Generator(device=dev); g.manual_seed(seed)
    acts = torch.randint(0, 7, (T, B, A), dtype=torch.int8, device=dev, generator=g)
    ctx = f"case {n_case}: {kind} {spec} B={B} staged={staged} lead={lead} mode={mode} seed={seed}"
    if os.environ.get("MGX_FUZZ_VERBOSE"):
        print(ctx, flush=True)
    if mode == "graph":
        K = min(T, int(r.integers(2, 2 * M + 2)))
        graph = env.capture_steps(acts[:K], auto_reset=True, sub_shards=int(r.choice([1, 1, 2])))
        for rep in range(2):
            graph.replay()
            for t in range(K):
                ref.step(acts[t], auto_reset=True)
        torch.cuda.synchronize()
        n_steps += 2 * K * B
    else:
        inject = kind == "bup" and r.random() < 0.5              # early ends: an agent is handed the target box now and then
        for t in range(T):
            if inject and t % 5 == 3:
                which = torch.from_numpy(np.random.default_rng(seed + t).random(B) < 0.3).to(dev)
                for e in (ref, env):
                    e.join()
                    e.agents[which, 0, 5] = 7
                    e.agents[which, 0, 6] = e.aux[which, 1]
            want = [x.clone() for x in ref.step(acts[t], auto_reset=True)] + [ref.was_reset.clone()]
            got = list(env.step(acts[t], auto_reset=True, **({"sub_shards": 2} if mode == "chains" and B >= 128 else {}))) + [env.was_reset]
            env.join()
            n_early += int((env.was_reset.bool() & ~env.truncated.bool()).sum())
            for k, (w, x) in enumerate(zip(want, got)):
                if not torch.equal(w, x):
                    print("MISMATCH step", t, "output", k, ctx); sys.exit(1)
        n_steps += T * B
    for f in FIELDS:
        if not torch.equal(getattr(ref, f), getattr(env, f)):
            print("MISMATCH state", f, ctx); sys.exit(1)
    if not torch.equal(ref._gen["gen_state"], env._gen["gen_state"]):
        print("MISMATCH generator state", ctx); sys.exit(1)
    env.check_errors()
    n_case += 1
print(f"fuzz ok: {n_case} random cases of staged generation == unstaged, {n_steps} env-steps, {n_early} early episode ends, {budget:.0f} s")
