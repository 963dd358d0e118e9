#!/usr/bin/env python3
"""Randomised soak: fused HIP step / rollout vs the CPU oracle on random specs, shapes and states (test infrastructure;
GPU box).  usage: python tools/fuzz_parity.py [seconds] [seed]      prints one line per case, exits 1 on the first mismatch."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multigrid_amd import BatchedMultiGridEnv, EnvSpec, layouts  # noqa: E402
from oracle import binding as ob  # noqa: E402
from tests import util  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
r = np.random.default_rng(seed0)
dev = "cuda:0"
t_end = time.time() + budget
n_case = n_steps = 0
while time.time() < t_end:
    V = int(r.choice([3, 5, 7, 7, 7, 9, 11, 13, 15]))
    W, H = int(r.integers(3, 41)), int(r.integers(3, 41))
    if r.random() < 0.1:
        W, H = int(r.integers(41, 100)), int(r.integers(41, 80))
    A = int(r.choice([1, 1, 2, 2, 3, 4, 4, 5, 7, 8, 12, 16, 32]))
    spec = EnvSpec(W, H, A, V, max_steps=int(r.integers(3, 60)),
                   see_through_walls=bool(r.random() < 0.2), allow_agent_overlap=bool(r.random() < 0.7),
                   joint_reward=bool(r.random() < 0.3),
                   success_termination_mode=str(r.choice(["any", "all"])),
                   failure_termination_mode=str(r.choice(["any", "all"])))
    B = int(r.choice([1, 2, 3, 7, 33, 64, 65, 255, 1000, 2049, 5000]))
    if r.random() < 0.06:                       # launches of more than 2048 wavefronts: the throughput instantiations
        W, H = int(r.integers(3, 13)), int(r.integers(3, 13))
        B = int(r.integers(30000, 150000))
    fixed = None
    if r.random() < 0.18:                       # the shape-specialised instantiations (mgx_fused.h: kShapes), at batches that pick them
        fixed = int(r.integers(0, 2))           # (the C5 shape: tests/test_full_size.py, at its full 32768 envs)
        V, W, H, A, B = [(7, 16, 16, 4, int(r.choice([5, 64, 1000, 4096, 8192]))), (7, 16, 16, 4, int(r.choice([8200, 12000, 16384])))][fixed]
        spec = EnvSpec(W, H, A, V, max_steps=int(r.integers(3, 60)), see_through_walls=spec.see_through_walls,
                       allow_agent_overlap=spec.allow_agent_overlap, joint_reward=spec.joint_reward,
                       success_termination_mode=spec.success_termination_mode, failure_termination_mode=spec.failure_termination_mode)
    if W * H * A * B > 4e7 and fixed is None:
        B = max(1, int(4e7 / (W * H * A)))
    T = int(r.integers(3, 14))
    seed = int(r.integers(0, 1 << 30))
    contents = float(r.choice([0.0, 0.0, 0.6]))       # boxes that hold things (Box.contains, include/mgx.h "BOX CONTENTS")
    st = util.random_state(spec, B, seed=seed, density=float(r.choice([0.0, 0.1, 0.3, 0.5])),
                           terminated_p=float(r.choice([0.0, 0.05, 0.3])), box_contents_p=contents)
    hook = r.random() < 0.15 and fixed is None
    if hook:                                    # a hook env kind on generated layouts: BlockedUnlockPickup, 2..4 agents
        A = int(r.integers(2, 5)); B = min(B, 300)
        if r.random() < 0.5:                    # (... half of them the shape with its own instantiation: 2 agents, 7x7 views)
            A, V = 2, 7
        spec = EnvSpec(11, 6, A, V if V <= 9 else 7, max_steps=int(r.integers(5, 80)), joint_reward=True,
                       env_kind="blockedunlockpickup", see_through_walls=spec.see_through_walls)
        lr = np.random.default_rng(seed)
        lay = [layouts.blockedunlockpickup_layout(6, A, lr, lr) for _ in range(B)]
        st = dict(grid=np.stack([x[0] for x in lay]), agents=np.stack([x[1] for x in lay]),
                  target=np.stack([layouts.make_aux("blockedunlockpickup", x[0], x[2]) for x in lay]),
                  rng=np.random.default_rng(seed + 5).integers(0, 2 ** 63, size=(B, 4), dtype=np.int64).astype(np.uint64) | np.uint64(1),
                  step_count=np.zeros(B, np.int32))
    first_env = int(r.integers(0, 1000))
    env = BatchedMultiGridEnv(spec, B, dev, first_env=first_env)
    env.load_state(st["grid"], st["agents"], st["rng"], st["target"], st["step_count"])
    roll = BatchedMultiGridEnv(spec, B, dev, first_env=first_env)
    roll.load_state(st["grid"], st["agents"], st["rng"], st["target"], st["step_count"])
    # the same state on COMPACT one-byte cells (include/mgx.h: MgxCell8), stepped beside it -- when nothing in it needs the 16-bit
    # cells (a box that holds something)
    comp = None
    if contents == 0.0 and r.random() < 0.6:
        import dataclasses
        comp = BatchedMultiGridEnv(dataclasses.replace(spec, cell_bytes=1), B, dev, first_env=first_env)
        comp.load_state(st["grid"], st["agents"], st["rng"], st["target"], st["step_count"])
    n_comp = (n_comp if n_case else 0) + (comp is not None)
    # ... and on the reference's own BYTE triples u8[B,H,W,3] (MgxSpec.cell_bytes = 3: packed in the step kernel's load phase,
    # written back as bytes; a box's content rides in the state byte)
    byt = None
    if r.random() < 0.5:
        import dataclasses
        byt = BatchedMultiGridEnv(dataclasses.replace(spec, cell_bytes=3), B, dev, first_env=first_env)
        byt.load_state(st["grid"], st["agents"], st["rng"], st["target"], st["step_count"])
    n_byt = (n_byt if n_case else 0) + (byt is not None)
    ref = {k: v.copy() for k, v in st.items()}
    sd = spec.as_dict()
    acts = np.stack([util.random_actions(B, A, seed=seed + 1 + t) for t in range(T)])
    if r.random() < 0.5:
        acts[r.random(acts.shape) < 0.1] = -1
    # half of the cases run with the fused auto-reset; the restarts are emulated here in numpy from the definition
    # (include/mgx.h: layout = (first_env + b + episode * 7919) mod K, step_count 0, episode + 1), then the oracle steps
    ar = bool(r.random() < 0.5) and not hook
    if ar:
        K = int(r.integers(1, 6))
        pool = util.random_state(spec, K, seed=seed + 77, terminated_p=0.0, density=0.3, box_contents_p=contents)
        env.set_layout_pool(pool["grid"], pool["agents"]); roll.set_layout_pool(pool["grid"], pool["agents"])
        if comp is not None:
            comp.set_layout_pool(pool["grid"], pool["agents"])
        if byt is not None:
            byt.set_layout_pool(pool["grid"], pool["agents"])
        episode = np.zeros(B, dtype=np.int64)
    rr = roll.rollout(torch.from_numpy(acts).to(dev), auto_reset=ar)
    ctx = f"case {n_case}: {spec} B={B} T={T} seed={seed} auto_reset={ar} fixed_shape={env.backend.launch_info(B).get('fixed_shape')}"
    n_fixed = (n_fixed if n_case else 0) + (1 if env.backend.launch_info(B).get("fixed_shape") else 0)
    for t in range(T):
        if ar:
            done = (ref["agents"][:, :, 4].min(axis=1) > 0) | (ref["step_count"] >= spec.max_steps)
            lay = (first_env + np.arange(B) + episode * 7919) % K
            ref["grid"][done] = pool["grid"][lay[done]]
            ref["agents"][done] = pool["agents"][lay[done]]
            ref["step_count"][done] = 0
            episode[done] += 1
        want = ob.step_batch(sd, ref["grid"], ref["agents"], ref["rng"], ref["step_count"], acts[t], ref["target"], nthreads=16)
        got = env.step(torch.from_numpy(acts[t]).to(dev), auto_reset=ar)
        if ar and (env.was_reset.cpu().numpy().astype(bool) != done).any():
            print("MISMATCH was_reset at step", t, ctx); sys.exit(1)
        for k, (g, w, key) in enumerate(zip(got, want, ("obs", "dir", "reward", "terminated", "truncated"))):
            if g.cpu().numpy().tobytes() != w.tobytes():
                print("MISMATCH step", t, key, ctx); sys.exit(1)
            if rr[key][t].cpu().numpy().tobytes() != w.tobytes():
                print("MISMATCH rollout step", t, key, ctx); sys.exit(1)
        if env.grid.cpu().numpy().tobytes() != ref["grid"].tobytes() or env.agents.cpu().numpy().tobytes() != ref["agents"].tobytes():
            print("MISMATCH state at step", t, ctx); sys.exit(1)
        if comp is not None:
            gotc = comp.step(torch.from_numpy(acts[t]).to(dev), auto_reset=ar)
            for g, w, key in zip(gotc, want, ("obs", "dir", "reward", "terminated", "truncated")):
                if g.cpu().numpy().tobytes() != w.tobytes():
                    print("MISMATCH compact cells, step", t, key, ctx); sys.exit(1)
            if comp.grid.cpu().numpy().tobytes() != ref["grid"].tobytes() or comp.agents.cpu().numpy().tobytes() != ref["agents"].tobytes():
                print("MISMATCH compact cells, state at step", t, ctx); sys.exit(1)
        if byt is not None:
            gotb = byt.step(torch.from_numpy(acts[t]).to(dev), auto_reset=ar)
            for g, w, key in zip(gotb, want, ("obs", "dir", "reward", "terminated", "truncated")):
                if g.cpu().numpy().tobytes() != w.tobytes():
                    print("MISMATCH byte grid, step", t, key, ctx); sys.exit(1)
            if byt.grid.cpu().numpy().tobytes() != ref["grid"].tobytes() or byt.agents.cpu().numpy().tobytes() != ref["agents"].tobytes():
                print("MISMATCH byte grid, state at step", t, ctx); sys.exit(1)
    if byt is not None:
        byt.check_errors()
    for e in (env, roll):
        if e.grid.cpu().numpy().tobytes() != ref["grid"].tobytes() or e.agents.cpu().numpy().tobytes() != ref["agents"].tobytes() \
                or e.step_count.cpu().numpy().tobytes() != ref["step_count"].tobytes() \
                or (A > 1 and e.rng.cpu().numpy().view(np.uint64).tobytes() != ref["rng"].tobytes()):
            print("MISMATCH final state", ctx); sys.exit(1)
    # the fused one-hot output of the final state == one-hot of the plain observation
    o_plain = env.gen_obs()[0].clone()
    if env.gen_obs(one_hot=True)[0].cpu().numpy().tobytes() != ob.one_hot(o_plain.cpu().numpy()).tobytes():
        print("MISMATCH one-hot gen_obs", ctx); sys.exit(1)
    env.check_errors(); roll.check_errors()
    if comp is not None:
        comp.check_errors()
    n_case += 1; n_steps += T * B
    del env, roll, comp
print(f"{n_case} cases, {n_steps} env-steps, {n_fixed} cases on a shape-specialised instantiation, {n_comp} also on compact cells, {n_byt} also on byte grids: clean")
from multigrid_amd import _lib  # noqa: E402
if hasattr(_lib.lib(), "mgx_debug_bounds_violations"):            # the checked build (MGX_LIBMGX=.../libmgx_chk.so)
    import ctypes
    v = (ctypes.c_int32 * 2)()
    assert _lib.lib().mgx_debug_bounds_violations(v) == 0
    print(f"bounds check: {v[0]} LDS accesses outside their wavefront's slice (last site {v[1]})")
    if v[0]:
        sys.exit(1)
print(f"fuzz ok: {n_case} random cases, {n_steps} env-steps, {budget:.0f} s, seed {seed0}")
