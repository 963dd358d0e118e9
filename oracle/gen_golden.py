#!/usr/bin/env python3
"""Generate golden vectors by running the REAL reference (ini/multigrid) in the build container.

TEST INFRASTRUCTURE -- not product code.  Runs only where /root/reference exists (never on the GPU box):

    python oracle/gen_golden.py            # rewrites tests/golden/*.npz

The reference's four absent third-party packages are satisfied by `oracle/standins/` (this repo's own
stand-ins; see its README).  Every array written here is *data produced by the reference*: initial state
tensors, the post-reset bit-generator state, the action script, and per step the reference's outputs
(`MultiGridEnv.step`, multigrid/base.py:303-346) and post-step state.  No reference source is copied.

Array conventions in the .npz files (reference layouts, narrowed to the smallest lossless dtype):
    grid0        (W,H,3)   Grid.state after reset (+ scenario edits)       multigrid/core/grid.py:54
    agents0      (A,9)     AgentState rows after reset                     multigrid/core/agent.py:222-232
    rng0         (4,) u64  PCG64 [state_hi, state_lo, inc_hi, inc_lo] after reset returned (SURVEY A.6 trap)
    actions      (T,A) i8  -1 = agent key absent from the actions dict     multigrid/base.py:403-404
    order        (T,A)     handle_actions' random visiting order           multigrid/base.py:396-399
    obs0         (A,v,v,3) reset observation images                        multigrid/base.py:295
    obs          (T,A,v,v,3), direction (T,A), reward (T,A) f64, terminated (T,A), truncated (T,)
    grid         (T,W,H,3), agents (T,A,9)  post-step state
    rng_final    (4,) u64
    spec_json    JSON: constructor-level parameters needed to rebuild the EnvSpec
    hook_order   (T,A) u8, only in the *_dictorder fixtures: the insertion order of the keys of the actions dict handed to
                 step() at step t (the RedBlueDoors / LockedHallway hooks iterate actions.items(), redbluedoors.py:176,
                 locked_hallway.py:210); every other fixture builds its dict in ascending agent order
"""
from __future__ import annotations

import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REFERENCE = os.environ.get("MGX_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True
sys.path.insert(0, REFERENCE)
sys.path.insert(0, os.path.join(HERE, "standins"))

import numpy as np  # noqa: E402

import multigrid.envs as ref_envs  # noqa: E402
from multigrid.core.world_object import Ball, Box, Door, Floor, Key, Lava, Wall  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")
_MAKE_COUNT = [0]
M64 = (1 << 64) - 1


def rng_words(gen) -> np.ndarray:
    st = gen.bit_generator.state["state"]
    s, inc = int(st["state"]), int(st["inc"])
    return np.array([s >> 64, s & M64, inc >> 64, inc & M64], dtype=np.uint64)


def clone_order(gen, n):
    """What handle_actions will draw next (base.py:396-399), without disturbing the env's generator."""
    if n == 1:
        return np.zeros(1, dtype=np.int64)
    bg = np.random.PCG64()
    bg.state = gen.bit_generator.state
    return np.random.Generator(bg).random(size=n).argsort()


def narrow(a):
    a = np.asarray(a)
    if a.dtype == np.bool_:
        return a.astype(np.uint8)
    if a.dtype.kind in "iu":
        lo, hi = (int(a.min()), int(a.max())) if a.size else (0, 0)
        if lo >= 0 and hi <= 255:
            return a.astype(np.uint8)
        if lo >= -128 and hi <= 127:
            return a.astype(np.int8)
        return a.astype(np.int32)
    return a


#: Box.contains (multigrid/core/world_object.py:574-605) is an attribute of the Python object: neither Grid.state nor the agent
#: rows show it.  The fixtures record it where the product's byte layouts carry it (include/mgx.h "BOX CONTENTS"): in the upper
#: bits of the box cell's STATE value, state | kind << 2 | colour << 5.  Every fixture without a filled box is unchanged by this.
_CONTENT_KIND = {"key": 1, "ball": 2, "goal": 3, "floor": 4, "lava": 5, "wall": 6, "door": 7}


def content_code(obj) -> int:
    inner = getattr(obj, "contains", None)
    if inner is None:
        return 0
    name = inner.type.value if hasattr(inner.type, "value") else str(inner.type)
    assert name in _CONTENT_KIND, f"box content {name!r} has no code"
    assert name != "door" or int(inner[2]) == 1, "a door in a box: closed and unlocked only"
    return _CONTENT_KIND[name] | (int(inner[1]) << 3)


def grid_with_contents(env) -> np.ndarray:
    """env.grid.state (W,H,3) + what the boxes lying on the grid hold."""
    st = env.grid.state.copy()
    for x, y in np.argwhere(st[..., 0] == 7):
        st[x, y, 2] |= content_code(env.grid.get(int(x), int(y))) << 2
    return st


def agents_with_contents(env) -> np.ndarray:
    """env.agent_states (A,9) + what the boxes being carried hold."""
    rows = np.asarray(env.agent_states).copy()
    for agent in env.agents:
        carried = agent.state.carrying
        if carried is not None and int(carried[0]) == 7:
            rows[agent.index, 8] |= content_code(carried) << 2
    return rows


def make_env(name, **kwargs):
    cls, cfg = ref_envs.CONFIGURATIONS[name]
    # pin the construction-time (layout) generator so regeneration is reproducible; see standins/gymnasium
    _MAKE_COUNT[0] += 1
    cls._default_seed = 0xC0FFEE + _MAKE_COUNT[0]
    return cls(**{**cfg, **kwargs})


def spec_of(env, kind, extra=None):
    d = dict(
        env_kind=kind,
        width=int(env.width), height=int(env.height), num_agents=int(env.num_agents),
        view_size=int(env.agents[0].view_size),
        see_through_walls=bool(env.agents[0].see_through_walls),
        allow_agent_overlap=bool(env.allow_agent_overlap),
        joint_reward=bool(env.joint_reward),
        success_termination_mode=str(env.success_termination_mode),
        failure_termination_mode=str(env.failure_termination_mode),
        max_steps=int(env.max_steps),
    )
    if kind == "blockedunlockpickup":
        d["target"] = [int(v) for v in np.asarray(env.obj)]
    if kind == "redbluedoors":
        d["blue_door"] = [int(v) for v in env.blue_door.cur_pos or door_pos(env, "blue")]
        d["red_door"] = [int(v) for v in env.red_door.cur_pos or door_pos(env, "red")]
    if kind == "lockedhallway":
        d["doors"] = sorted([int(x), int(y)] for x, y in np.argwhere(env.grid.state[..., 0] == 4))
        d["num_room_keys"] = len(env.rooms)          # rooms keyed by colour: what ends the episode (locked_hallway.py:222-225)
    d.update(extra or {})
    return d


def door_pos(env, color):
    ci = ["red", "green", "blue", "purple", "yellow", "grey"].index(color)
    st = env.grid.state
    (x, y), = np.argwhere((st[..., 0] == 4) & (st[..., 1] == ci))
    return int(x), int(y)


def record(fname, env, kind, seed, T, action_rng, p_missing=0.0, edit=None, script=None, note="", dict_orders=None):
    obs0, _ = env.reset(seed=seed)
    if edit is not None:
        edit(env)
        obs0 = env.gen_obs()
    A = env.num_agents
    rec = dict(
        grid0=narrow(grid_with_contents(env)),
        agents0=narrow(agents_with_contents(env)),
        rng0=rng_words(env.np_random),
        obs0=narrow(np.stack([obs0[i]["image"] for i in range(A)])),
        dir0=narrow(np.array([int(obs0[i]["direction"]) for i in range(A)])),
    )
    if script is not None:
        actions = np.asarray(script, dtype=np.int8)
        T = len(actions)
    else:
        actions = action_rng.integers(0, 7, size=(T, A)).astype(np.int8)
        if p_missing > 0:
            actions[action_rng.random((T, A)) < p_missing] = -1
    keys = ("order", "obs", "direction", "reward", "terminated", "truncated", "grid", "agents")
    log = {k: [] for k in keys}
    for t in range(T):
        log["order"].append(clone_order(env.np_random, A))
        key_order = range(A) if dict_orders is None else [int(i) for i in dict_orders[t]]
        act = {i: int(actions[t, i]) for i in key_order if actions[t, i] >= 0}
        obs, rew, term, trunc, _ = env.step(act)
        log["obs"].append(np.stack([obs[i]["image"] for i in range(A)]))
        log["direction"].append([int(obs[i]["direction"]) for i in range(A)])
        log["reward"].append([float(rew[i]) for i in range(A)])
        log["terminated"].append([bool(term[i]) for i in range(A)])
        log["truncated"].append(bool(trunc[0]))
        assert len(set(bool(trunc[i]) for i in range(A))) == 1
        log["grid"].append(grid_with_contents(env))
        log["agents"].append(agents_with_contents(env))
    rec["actions"] = actions
    if dict_orders is not None:
        rec["hook_order"] = np.asarray(dict_orders, dtype=np.uint8)[:T]
    for k in keys:
        arr = np.asarray(log[k])
        rec[k] = arr.astype(np.float64) if k == "reward" else narrow(arr)
    rec["rng_final"] = rng_words(env.np_random)
    rec["spec_json"] = np.array(json.dumps(spec_of(env, kind, dict(seed=seed, note=note))))
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, fname + ".npz")
    np.savez_compressed(path, **rec)
    n_succ = int((rec["reward"] > 0).any(axis=1).sum())
    print(f"{fname:34s} T={T:4d} A={A:2d} grid={env.width}x{env.height} "
          f"success-steps={n_succ:3d} terminated-end={rec['terminated'][-1].tolist()} "
          f"{os.path.getsize(path) / 1024:.1f} KiB")


def sprinkle(density, seed, doors=True):
    """Scenario edit: scatter every object type over interior cells not under an agent (uses the
    reference's own WorldObj classes and Grid.set, multigrid/core/grid.py:78-100)."""
    colors = ["red", "green", "blue", "purple", "yellow", "grey"]

    def edit(env):
        r = np.random.default_rng(seed)
        agent_cells = {tuple(int(v) for v in a.state.pos) for a in env.agents}
        for x in range(1, env.width - 1):
            for y in range(1, env.height - 1):
                if (x, y) in agent_cells or env.grid.get(x, y) is not None:
                    continue
                if r.random() >= density:
                    continue
                c = colors[int(r.integers(6))]
                k = int(r.integers(9 if doors else 6))
                obj = [lambda: Lava(), lambda: Floor(c), lambda: Key(c), lambda: Ball(c), lambda: Box(c),
                       lambda: Wall(), lambda: Door(c), lambda: Door(c, is_locked=True),
                       lambda: Door(c, is_open=True)][k]()
                env.grid.set(x, y, obj)
    return edit


def main():
    # C1: the reference's own CPU-runnable case (BASELINE.json configs[0])
    record("empty8_a2_seed0", make_env("MultiGrid-Empty-8x8-v0", agents=2), "empty", 0, 256,
           np.random.default_rng(0), note="C1 plumbing case; actions default_rng(0).integers(0,7,(T,A))")
    # C2/C4 shape: Empty-16x16, 4 agents, view 7
    for seed in (1, 7):
        record(f"empty16_a4_seed{seed}", make_env("MultiGrid-Empty-16x16-v0", agents=4), "empty", seed, 300,
               np.random.default_rng(100 + seed), note="C2/C4 shape")
    # biased towards 'forward' so agents reach the goal: exercises on_success ('any') + reward value
    def fwd_script(T, A, seed):
        r = np.random.default_rng(seed)
        return r.choice([0, 1, 2, 2, 2, 2, 2, 5], size=(T, A)).astype(np.int8)
    record("empty6_a3_goal", make_env("MultiGrid-Empty-6x6-v0", agents=3), "empty", 3, None, None,
           script=fwd_script(120, 3, 5), note="goal reached under success 'any'; steps after termination")
    record("empty6_a3_goal_all_joint",
           make_env("MultiGrid-Empty-6x6-v0", agents=3, success_termination_mode="all", joint_reward=True),
           "empty", 4, None, None, script=fwd_script(200, 3, 6), note="success 'all' + joint reward")
    # every object type, doors in all 3 states, 5% missing actions
    record("empty16_a4_objects", make_env("MultiGrid-Empty-16x16-v0", agents=4), "empty", 11, 400,
           np.random.default_rng(211), p_missing=0.05, edit=sprinkle(0.30, 5), note="all object types")
    record("empty8_a3_dense", make_env("MultiGrid-Empty-8x8-v0", agents=3), "empty", 12, 400,
           np.random.default_rng(212), p_missing=0.05, edit=sprinkle(0.45, 6), note="dense pickup/drop/toggle")
    # lava with failure 'any' and 'all'
    record("empty8_a3_lava_any",
           make_env("MultiGrid-Empty-8x8-v0", agents=3, failure_termination_mode="any"), "empty", 13, None, None,
           edit=sprinkle(0.25, 7, doors=False), script=fwd_script(150, 3, 8), note="lava, failure 'any'")
    record("empty8_a3_lava_all", make_env("MultiGrid-Empty-8x8-v0", agents=3), "empty", 14, None, None,
           edit=sprinkle(0.25, 9, doors=False), script=fwd_script(150, 3, 10), note="lava, failure 'all'")
    # no agent overlap, view 5, random starts
    record("emptyrand6_a3_nooverlap_v5",
           make_env("MultiGrid-Empty-Random-6x6-v0", agents=3, allow_agent_overlap=False, agent_view_size=5),
           "empty", 21, 300, np.random.default_rng(221), p_missing=0.05, note="allow_agent_overlap=False, v=5")
    # single agent (no RNG draw, no overlay), see_through_walls
    record("empty8_a1_seethrough", make_env("MultiGrid-Empty-8x8-v0", agents=1, see_through_walls=True),
           "empty", 22, 200, np.random.default_rng(222), edit=sprinkle(0.3, 11), note="A=1, see_through_walls")
    record("empty8_a1_walls", make_env("MultiGrid-Empty-8x8-v0", agents=1),
           "empty", 23, 200, np.random.default_rng(223), edit=sprinkle(0.3, 12), note="A=1, occlusion")
    # 5 agents, v=9, success all / failure any / joint reward / random starts
    record("empty16_a5_v9_modes",
           make_env("MultiGrid-Empty-16x16-v0", agents=5, agent_view_size=9, agent_start_pos=None,
                    success_termination_mode="all", failure_termination_mode="any", joint_reward=True),
           "empty", 31, 300, np.random.default_rng(231), p_missing=0.05, edit=sprinkle(0.15, 13),
           note="v=9, modes")
    # tiny view / large view
    record("empty8_a2_v3", make_env("MultiGrid-Empty-8x8-v0", agents=2, agent_view_size=3), "empty", 32, 150,
           np.random.default_rng(232), edit=sprinkle(0.3, 14), note="v=3")
    record("empty16_a2_v11", make_env("MultiGrid-Empty-16x16-v0", agents=2, agent_view_size=11,
                                      agent_start_pos=None), "empty", 33, 150,
           np.random.default_rng(233), edit=sprinkle(0.25, 15), note="v=11")
    # max_steps truncation boundary
    record("empty5_a2_trunc", make_env("MultiGrid-Empty-5x5-v0", agents=2, max_steps=20), "empty", 34, 40,
           np.random.default_rng(234), note="truncation at step 20, stepping beyond")
    # C5 shape: 64x64, 16 agents, view 9
    record("empty64_a16_v9", make_env("MultiGrid-Empty-8x8-v0", size=64, agents=16, agent_view_size=9,
                                      agent_start_pos=None), "empty", 41, 60,
           np.random.default_rng(241), edit=sprinkle(0.08, 16), note="C5 shape")
    # C3: BlockedUnlockPickup (RoomGrid draws from the seeded stream inside reset -> rng0 snapshot matters)
    for A, seed in ((2, 51), (3, 52)):
        record(f"bup_a{A}_seed{seed}", make_env("MultiGrid-BlockedUnlockPickup-v0", agents=A),
               "blockedunlockpickup", seed, 300, np.random.default_rng(250 + A), p_missing=0.03, note="C3")

    # C3 with the post-step success hook firing: teleport agent 0 next to the target box, facing it
    def next_to_box(env):
        bx, by = env.obj.cur_pos
        for d, (dx, dy) in enumerate([(1, 0), (0, 1), (-1, 0), (0, -1)]):
            x, y = bx - dx, by - dy
            if env.grid.get(x, y) is None:
                env.agents[0].state.pos = (x, y)
                env.agents[0].state.dir = d
                return
        raise RuntimeError("box is boxed in")
    hook = np.array([[6, 2]] * 3 + [[3, 0]] + [[2, 1], [4, 2], [3, 5], [5, 3], [0, 4]] * 4, dtype=np.int8)
    record("bup_a2_hook", make_env("MultiGrid-BlockedUnlockPickup-v0", agents=2), "blockedunlockpickup",
           53, None, None, edit=next_to_box, script=hook, note="post-step hook: carrying target box -> success")

    # scripted key / locked-door protocol (world_object.py:458-474): wrong key, no key, unlock, close, reopen
    def key_and_doors(env):
        env.grid.set(2, 1, Key("blue"))
        env.grid.set(3, 1, Door("blue", is_locked=True))
        env.grid.set(2, 2, Door("red", is_locked=True))
        for y in range(2, 7):
            env.grid.set(3, y, Wall())
    script = [[3, 6], [2, 2], [6, 5], [1, 6], [5, 6], [0, 6], [5, 6], [6, 5], [6, 5], [2, 2], [2, 6], [4, 6],
              [6, 5]]
    script += np.random.default_rng(77).integers(0, 7, size=(80, 2)).tolist()
    record("empty8_a2_unlock", make_env("MultiGrid-Empty-8x8-v0", agents=2), "empty", 61, None, None,
           edit=key_and_doors, script=script, note="scripted unlock / wrong key / close / reopen")

    record_layouts()
    record_wrappers()
    record_hook_envs()
    record_dict_order()
    record_custom()          # (last: every make_env() before it keeps the construction seed it always had)
    record_custom_steps()    # (round 5, behind everything older for the same reason)
    record_box_rollout()
    import custom_envs       # (round 6, likewise: the RoomGrid subclass of tests/custom_envs.py)
    record_custom(custom_envs.CASES_R6)
    record_custom_steps(custom_envs.STEP_CASES_R6)


def face(env, i, target_xy, carrying=None):
    """Teleport agent i next to `target_xy`, facing it (first free side)."""
    tx, ty = target_xy
    for d, (dx, dy) in enumerate([(1, 0), (0, 1), (-1, 0), (0, -1)]):
        x, y = tx - dx, ty - dy
        if 0 < x < env.width - 1 and 0 < y < env.height - 1 and env.grid.get(x, y) is None:
            env.agents[i].state.pos = (x, y)
            env.agents[i].state.dir = d
            if carrying is not None:
                env.agents[i].state.carrying = carrying
            return
    raise RuntimeError("no free side")


def record_hook_envs():
    """Section 8f-4: RedBlueDoors (envs/redbluedoors.py:170-187, incl. the stale-grid quirk Q9), LockedHallway
    (envs/locked_hallway.py:203-227) and Playground (no hook; RoomGrid layout) -- rollouts of the real reference."""
    from multigrid.core.world_object import Key
    T, L, R, F, P, D, G, N = 5, 0, 1, 2, 3, 4, 5, 6   # toggle, left, right, forward, pickup, drop, toggle(again), done

    def rbd_success(env):
        face(env, 0, door_pos(env, "red")); face(env, 1, door_pos(env, "blue"))
    record("rbd_a2_success", make_env("MultiGrid-RedBlueDoors-8x8-v0", agents=2), "redbluedoors", 71, None, None,
           edit=rbd_success, script=[[T, N], [N, T], [N, T], [T, T], [F, F], [N, T]],
           note="red then blue: success; hook keeps firing for terminated agents")

    def rbd_failure(env):
        face(env, 1, door_pos(env, "blue"))
        bx, by = door_pos(env, "blue")
        env.agents[0].state.pos = (bx - 1, by); env.agents[0].state.dir = 0
    for mode in ("any", "all"):
        record(f"rbd_a2_failure_{mode}",
               make_env("MultiGrid-RedBlueDoors-8x8-v0", agents=2, failure_termination_mode=mode), "redbluedoors", 72,
               None, None, edit=rbd_failure,
               script=[[N, T], [F, N], [T, N], [F, T], [T, N], [F, F], [T, T], [F, F], [L, T], [T, N]],
               note="blue first: failure; the blue door object closes while grid.state keeps saying open (Q9)")
    record("rbd_a3_random", make_env("MultiGrid-RedBlueDoors-6x6-v0", agents=3, failure_termination_mode="all"),
           "redbluedoors", 73, 400, np.random.default_rng(373), p_missing=0.05, note="random rollout")

    def lh_unlock(env):
        doors = sorted((int(x), int(y)) for x, y in np.argwhere(env.grid.state[..., 0] == 4))
        colors = ["red", "green", "blue", "purple", "yellow", "grey"]
        for i, (x, y) in enumerate(doors[:env.num_agents]):
            face(env, i, (x, y), carrying=Key(colors[int(env.grid.state[x, y, 1])]))
    for name, A, jr in (("MultiGrid-LockedHallway-2Rooms-v0", 2, True), ("MultiGrid-LockedHallway-2Rooms-v0", 2, False),
                        ("MultiGrid-LockedHallway-6Rooms-v0", 3, True)):
        rooms = name.split("-")[2]
        record(f"lh_{rooms.lower()}_a{A}_{'joint' if jr else 'own'}", make_env(name, agents=A, joint_reward=jr),
               "lockedhallway", 74 + A, None, None, edit=lh_unlock,
               script=[[N] * A, [T] + [N] * (A - 1), [T] * A, [T] * A, [F] * A, [T] * A, [L] * A, [T] * A] +
                      np.random.default_rng(5).integers(0, 7, size=(30, A)).tolist(),
               note="unlock with key: reward accumulates; all doors unlocked -> returned terminations only")
    record("lh_4rooms_a2_random", make_env("MultiGrid-LockedHallway-4Rooms-v0", agents=2), "lockedhallway", 77, 300,
           np.random.default_rng(377), p_missing=0.05, note="random rollout")
    record("playground_a3", make_env("MultiGrid-Playground-v0", agents=3), "empty", 78, 100,
           np.random.default_rng(378), p_missing=0.05, note="Playground: no hook, 19x19 RoomGrid layout, max_steps 100")
    # more rooms than colours (not a registered id: the class called directly): doors repeat colours, len(self.rooms) < doors
    _MAKE_COUNT[0] += 1
    ref_envs.LockedHallwayEnv._default_seed = 0xC0FFEE + _MAKE_COUNT[0]
    record("lh_8rooms_a8_joint", ref_envs.LockedHallwayEnv(num_rooms=8, agents=8, joint_reward=True), "lockedhallway", 81,
           None, None, edit=lh_unlock,
           script=[[N] * 8, [T] + [N] * 7, [T] * 8, [T] * 8, [F] * 8, [T] * 8, [L] * 8, [T] * 8] +
                  np.random.default_rng(6).integers(0, 7, size=(30, 8)).tolist(),
           note="8 rooms, 6 colours: rooms dict keyed by colour; every agent unlocks a door")


def record_dict_order():
    """The env hooks visit the agents in the insertion order of the caller's dict (redbluedoors.py:176, locked_hallway.py:210).
    Cases where that order decides the result, plus random rollouts with a random key order every step."""
    from multigrid.core.world_object import Key
    T, L, R, F, P, D, N = 5, 0, 1, 2, 3, 4, 6

    # RedBlueDoors, failure 'all', three agents stacked in front of the CLOSED blue door, all toggle in one step: the door
    # goes closed -> open -> closed -> open in handle_actions, then the hook fails the FIRST toggler in dict order only (it
    # closes the blue door object, so the next ones find it closed)
    def rbd_three(env):
        face(env, 0, door_pos(env, "blue"))
        for i in (1, 2):
            env.agents[i].state.pos = env.agents[0].state.pos
            env.agents[i].state.dir = env.agents[0].state.dir
    for tag, orders in (("rev", [[2, 1, 0]] * 6), ("mid", [[1, 2, 0], [1, 0, 2], [2, 0, 1], [0, 2, 1], [1, 2, 0], [2, 1, 0]])):
        record(f"rbd_a3_dictorder_{tag}",
               make_env("MultiGrid-RedBlueDoors-8x8-v0", agents=3, failure_termination_mode="all"), "redbluedoors", 91, None, None,
               edit=rbd_three, script=[[T, T, T], [N, N, N], [T, T, T], [T, N, T], [T, T, N], [T, T, T]], dict_orders=orders,
               note="three togglers of the blue door in one step: the first in DICT order fails (failure 'all')")

    # LockedHallway, own rewards: agent 0 (with the key) and agent 1 face the same locked door from the hallway and both toggle:
    # whoever comes first in dict order is paid for the unlock
    def lh_two(env):
        doors = sorted((int(x), int(y)) for x, y in np.argwhere(env.grid.state[..., 0] == 4))
        colors = ["red", "green", "blue", "purple", "yellow", "grey"]
        x, y = doors[0]
        face(env, 0, (x, y), carrying=Key(colors[int(env.grid.state[x, y, 1])]))
        env.agents[1].state.pos = env.agents[0].state.pos
        env.agents[1].state.dir = env.agents[0].state.dir
    for tag, orders in (("rev", [[1, 0]] * 8), ("fwd", [[0, 1]] * 8)):
        record(f"lh_2rooms_a2_dictorder_{tag}", make_env("MultiGrid-LockedHallway-2Rooms-v0", agents=2, joint_reward=False),
               "lockedhallway", 92, None, None, edit=lh_two,
               script=[[N, N], [T, T], [T, T], [T, N], [N, T], [T, T], [F, F], [T, T]], dict_orders=orders,
               note="two togglers of one door in one step: the first in DICT order is paid (joint_reward=False)")

    # random rollouts, a random permutation of the keys every step, agents kept near the doors by a biased action mix
    def perm_orders(T_, A, seed):
        r = np.random.default_rng(seed)
        return np.stack([r.permutation(A) for _ in range(T_)]).astype(np.uint8)

    def rbd_near(env):
        face(env, 0, door_pos(env, "blue")); face(env, 1, door_pos(env, "red"))
        env.agents[2].state.pos = env.agents[0].state.pos; env.agents[2].state.dir = env.agents[0].state.dir
    tog = np.random.default_rng(94).choice([T, T, T, L, R, F, N], size=(200, 3)).astype(np.int8)
    record("rbd_a3_random_dictorder",
           make_env("MultiGrid-RedBlueDoors-6x6-v0", agents=3, failure_termination_mode="all", success_termination_mode="all"),
           "redbluedoors", 93, None, None, edit=rbd_near, script=tog, dict_orders=perm_orders(200, 3, 95),
           note="toggle-heavy random rollout, random dict order every step")
    record("lh_4rooms_a3_random_dictorder", make_env("MultiGrid-LockedHallway-4Rooms-v0", agents=3, joint_reward=False),
           "lockedhallway", 96, 300, np.random.default_rng(397), p_missing=0.05, dict_orders=perm_orders(300, 3, 98),
           note="random rollout, random dict order every step")

    # more than 6 rooms with an EVEN room size -- the doors sit at (top + bottom) // 2 = row (rs-1) + (rs-1)//2
    def lh_unlock(env):
        doors = sorted((int(x), int(y)) for x, y in np.argwhere(env.grid.state[..., 0] == 4))
        colors = ["red", "green", "blue", "purple", "yellow", "grey"]
        for i, (x, y) in enumerate(doors[:env.num_agents]):
            face(env, i, (x, y), carrying=Key(colors[int(env.grid.state[x, y, 1])]))
    for rs in (4, 6):
        _MAKE_COUNT[0] += 1
        ref_envs.LockedHallwayEnv._default_seed = 0xC0FFEE + _MAKE_COUNT[0]
        record(f"lh_8rooms_rs{rs}_a4", ref_envs.LockedHallwayEnv(num_rooms=8, room_size=rs, agents=4, joint_reward=True),
               "lockedhallway", 99, None, None, edit=lh_unlock,
               script=[[N] * 4, [T] + [N] * 3, [T] * 4, [T] * 4, [F] * 4, [T] * 4, [L] * 4, [T] * 4] +
                      np.random.default_rng(7).integers(0, 7, size=(30, 4)).tolist(),
               note=f"8 rooms of even size {rs}: mid-wall doors at (top + bottom) // 2")


def record_wrappers():
    """OneHotObsWrapper / FullyObsWrapper outputs of the real reference (multigrid/wrappers.py:17-190) along a short
    rollout with every object type present."""
    from multigrid.wrappers import FullyObsWrapper, OneHotObsWrapper
    for fname, kw in (("wrappers_empty8_a3", dict(agents=3)), ("wrappers_empty16_a4_v5", dict(size=16, agents=4, agent_view_size=5))):
        base = make_env("MultiGrid-Empty-8x8-v0", **kw)
        base.reset(seed=9)
        sprinkle(0.3, 21)(base)
        A = base.num_agents
        oh, fo = OneHotObsWrapper(base), FullyObsWrapper(base)
        ar = np.random.default_rng(4)
        rec = dict(obs=[], one_hot=[], full=[], grid=[], agents=[])
        for t in range(40):
            act = {i: int(a) for i, a in enumerate(ar.integers(0, 7, size=A))}
            base.step(act)
            raw = base.gen_obs()
            rec["obs"].append(np.stack([raw[i]["image"] for i in range(A)]))
            o1 = oh.observation(base.gen_obs())
            rec["one_hot"].append(np.stack([o1[i]["image"] for i in range(A)]))
            o2 = fo.observation(base.gen_obs())
            assert all(o2[i]["image"] is o2[0]["image"] for i in range(A))
            rec["full"].append(np.array(o2[0]["image"]))
            rec["grid"].append(base.grid.state.copy()); rec["agents"].append(np.asarray(base.agent_states).copy())
        out = {k: narrow(np.asarray(v)) for k, v in rec.items()}
        out["spec_json"] = np.array(json.dumps(spec_of_noreset(base, "empty")))
        path = os.path.join(OUT, fname + ".npz")
        np.savez_compressed(path, **out)
        print(f"{fname:34s} one_hot{out['one_hot'].shape} full{out['full'].shape} {os.path.getsize(path) / 1024:.1f} KiB")


def record_layouts():
    """Layout / reset parity fixtures: a sequence of resets (seeded and unseeded, with steps in between) of one
    env object whose construction-time generator is known.  Pins multigrid_amd/layouts.py + MultiGridEnv.reset
    against multigrid/base.py:250-301, envs/empty.py:151-170, envs/blockedunlockpickup.py:142-164."""
    cases = [
        ("layout_bup_a2", "MultiGrid-BlockedUnlockPickup-v0", dict(agents=2), "blockedunlockpickup"),
        ("layout_bup_a3", "MultiGrid-BlockedUnlockPickup-v0", dict(agents=3), "blockedunlockpickup"),
        ("layout_emptyrandom6_a3", "MultiGrid-Empty-Random-6x6-v0", dict(agents=3), "empty"),
        ("layout_empty8_a2", "MultiGrid-Empty-8x8-v0", dict(agents=2), "empty"),
        ("layout_rbd8_a2", "MultiGrid-RedBlueDoors-8x8-v0", dict(agents=2), "redbluedoors"),
        ("layout_lh4_a2", "MultiGrid-LockedHallway-4Rooms-v0", dict(agents=2), "lockedhallway"),
        ("layout_playground_a2", "MultiGrid-Playground-v0", dict(agents=2), "empty"),
    ]
    for fname, name, kw, kind in cases:
        env = make_env(name, **kw)
        construct_seed = type(env)._default_seed
        A = env.num_agents
        reset_seeds = [5, -1, -1, 17, -1, 123456789, -1, -1]
        ar = np.random.default_rng(99)
        rec = dict(construct_seed=np.array(construct_seed), reset_seeds=np.array(reset_seeds),
                   spec_json=np.array(json.dumps(spec_of_noreset(env, kind))))
        grids, agents, rngs, obs0s, missions, acts_all, targets = [], [], [], [], [], [], []
        for k, sd in enumerate(reset_seeds):
            obs, _ = env.reset(seed=None if sd < 0 else sd)
            grids.append(env.grid.state.copy()); agents.append(np.asarray(env.agent_states).copy())
            rngs.append(rng_words(env.np_random))
            obs0s.append(np.stack([obs[i]["image"] for i in range(A)]))
            # mission_space.seed(None) draws OS entropy (base.py:272): only seeded resets have a reproducible mission
            missions.append(str(obs[0]["mission"]) if sd >= 0 else "<unseeded>")
            targets.append([int(v) for v in np.asarray(env.obj)] if kind == "blockedunlockpickup" else [0, 0, 0])
            acts = ar.integers(0, 7, size=(5, A)).astype(np.int8)
            acts_all.append(acts)
            for t in range(5):
                env.step({i: int(acts[t, i]) for i in range(A)})
        rec.update(grid0=narrow(np.asarray(grids)), agents0=narrow(np.asarray(agents)), rng0=np.asarray(rngs),
                   obs0=narrow(np.asarray(obs0s)), missions=np.array(missions), actions=np.asarray(acts_all),
                   targets=narrow(np.asarray(targets)))
        path = os.path.join(OUT, fname + ".npz")
        np.savez_compressed(path, **rec)
        print(f"{fname:34s} resets={len(reset_seeds)} missions={sorted(set(missions))[:2]} "
              f"{os.path.getsize(path) / 1024:.1f} KiB")


def record_custom(cases=None):
    """User-defined envs written against the reference's extension point, `_gen_grid` + `put_obj` / `place_obj` / `place_agent` /
    `Grid.wall_rect / horz_wall / vert_wall` / the WorldObj classes (multigrid/base.py:229-247, 604-697; core/grid.py:78-195;
    core/world_object.py:279-616).  The class bodies are this repo's own (tests/custom_envs.py) and run here over the REAL
    reference; the same bodies run over multigrid_amd in tests/test_custom_envs.py.  Same layout as the reset fixtures above,
    plus the five steps after every reset with their outputs."""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import custom_envs
    classes = custom_envs.define(custom_envs.multigrid_namespace())
    for fname, (cname, kw) in (custom_envs.CASES if cases is None else cases).items():
        cls = classes[cname]
        _MAKE_COUNT[0] += 1
        cls._default_seed = 0xC0FFEE + _MAKE_COUNT[0]
        env = cls(**kw)
        construct_seed = cls._default_seed
        A = env.num_agents
        reset_seeds = [3, -1, 44, -1, -1, 2024]
        ar = np.random.default_rng(7)
        rec = dict(construct_seed=np.array(construct_seed), reset_seeds=np.array(reset_seeds),
                   spec_json=np.array(json.dumps(spec_of_noreset(env, "empty"))))
        grids, agents, rngs, obs0s, acts_all, obs_s, rew_s, term_s = [], [], [], [], [], [], [], []
        for k, sd in enumerate(reset_seeds):
            obs, _ = env.reset(seed=None if sd < 0 else sd)
            grids.append(env.grid.state.copy()); agents.append(np.asarray(env.agent_states).copy())
            rngs.append(rng_words(env.np_random))
            obs0s.append(np.stack([obs[i]["image"] for i in range(A)]))
            acts = ar.choice([0, 1, 2, 2, 2, 3, 4, 5, 6], size=(5, A)).astype(np.int8)
            acts_all.append(acts)
            o_k, r_k, t_k = [], [], []
            for t in range(5):
                o, r, tm, tr, _ = env.step({i: int(acts[t, i]) for i in range(A)})
                o_k.append(np.stack([o[i]["image"] for i in range(A)]))
                r_k.append([float(r[i]) for i in range(A)]); t_k.append([bool(tm[i]) for i in range(A)])
            obs_s.append(o_k); rew_s.append(r_k); term_s.append(t_k)
        rec.update(grid0=narrow(np.asarray(grids)), agents0=narrow(np.asarray(agents)), rng0=np.asarray(rngs),
                   obs0=narrow(np.asarray(obs0s)), actions=np.asarray(acts_all), obs_steps=narrow(np.asarray(obs_s)),
                   reward_steps=np.asarray(rew_s, dtype=np.float64), terminated_steps=narrow(np.asarray(term_s)))
        path = os.path.join(OUT, fname + ".npz")
        np.savez_compressed(path, **rec)
        ntypes = sorted(set(int(v) for v in np.asarray(grids)[..., 0].ravel()))
        print(f"{fname:34s} resets={len(reset_seeds)} cell types seen={ntypes} {os.path.getsize(path) / 1024:.1f} KiB")


def record_custom_steps(cases=None):
    """User-defined envs stepped for whole episodes over the REAL reference (tests/custom_envs.py: STEP_CASES): boxes that hold
    things (Box.contains, world_object.py:574-605) and a `step` override that ends episodes through `on_success` / `on_failure`
    (base.py:478-532) the way the reference's own envs do.  Per reset: the initial state, then every step's actions, outputs and
    post-step state (box contents folded into the state values, see grid_with_contents)."""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import custom_envs
    classes = custom_envs.define(custom_envs.multigrid_namespace())
    for fname, (cname, kw, T) in (custom_envs.STEP_CASES if cases is None else cases).items():
        cls = classes[cname]
        _MAKE_COUNT[0] += 1
        cls._default_seed = 0xC0FFEE + _MAKE_COUNT[0]
        env = cls(**kw)
        A = env.num_agents
        reset_seeds = [11, -1, 5, 77]
        ar = np.random.default_rng(13)
        rec = dict(construct_seed=np.array(cls._default_seed), reset_seeds=np.array(reset_seeds),
                   spec_json=np.array(json.dumps(spec_of_noreset(env, "empty"))))
        keys = ("grid0", "agents0", "rng0", "obs0", "actions", "obs", "reward", "terminated", "truncated", "grid", "agents")
        log = {k: [] for k in keys}
        events = dict(success=0, failure=0, opened=0)
        for sd in reset_seeds:
            obs, _ = env.reset(seed=None if sd < 0 else sd)
            log["grid0"].append(grid_with_contents(env)); log["agents0"].append(agents_with_contents(env))
            log["rng0"].append(rng_words(env.np_random))
            log["obs0"].append(np.stack([obs[i]["image"] for i in range(A)]))
            # walk, pick up, drop and toggle a lot: boxes get opened and carried, balls fetched, the trap door opened
            acts = ar.choice(7, size=(T, A), p=[0.12, 0.12, 0.36, 0.12, 0.08, 0.17, 0.03]).astype(np.int8)
            acts[ar.random((T, A)) < 0.04] = -1
            if cname == "BoxTreasureEnv":
                acts[0, 0] = 5                                   # agent 0 starts facing the key box: open it, take the key
                acts[1, 0] = 3
            ep = {k: [] for k in ("obs", "reward", "terminated", "truncated", "grid", "agents")}
            for t in range(T):
                forced = custom_envs.intervene(cname, env, t)          # (tests/custom_envs.py: the replaying tests make the same call)
                for i, a_i in (forced or {}).items():
                    acts[t, i] = a_i
                nbox = int((env.grid.state[..., 0] == 7).sum())
                o, r, tm, tr, _ = env.step({i: int(acts[t, i]) for i in range(A) if acts[t, i] >= 0})
                ep["obs"].append(np.stack([o[i]["image"] for i in range(A)]))
                ep["reward"].append([float(r[i]) for i in range(A)]); ep["terminated"].append([bool(tm[i]) for i in range(A)])
                ep["truncated"].append(bool(tr[0]))
                ep["grid"].append(grid_with_contents(env)); ep["agents"].append(agents_with_contents(env))
                events["success"] += any(float(r[i]) > 0 for i in range(A))
                events["failure"] += any(bool(tm[i]) and float(r[i]) == 0 for i in range(A)) and not any(float(r[i]) > 0 for i in range(A))
                carried = sum(a.state.carrying is not None and int(a.state.carrying[0]) == 7 for a in env.agents)
                events["opened"] += int(nbox > int((env.grid.state[..., 0] == 7).sum()) + carried)
            log["actions"].append(acts)
            for k, v in ep.items():
                log[k].append(v)
        for k in keys:
            arr = np.asarray(log[k])
            rec[k] = arr.astype(np.float64) if k == "reward" else (arr if k in ("rng0", "actions") else narrow(arr))
        path = os.path.join(OUT, fname + ".npz")
        np.savez_compressed(path, **rec)
        print(f"{fname:34s} episodes={len(reset_seeds)} x {T} steps events={events} {os.path.getsize(path) / 1024:.1f} KiB")


def record_box_rollout():
    """A rollout fixture in the format of `record` from the BoxTreasureEnv of tests/custom_envs.py: replayed by every golden test
    (oracle, host rules, the HIP kernels through the C ABI) like the fixtures of the reference's own envs.  Agent 0 opens the box
    that holds the purple key, takes the key, walks to the locked purple door and unlocks it (scripted); the others act at random
    (toggle / pickup / drop heavy), so boxes are opened, carried and opened elsewhere."""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import custom_envs
    cls = custom_envs.define(custom_envs.multigrid_namespace())["BoxTreasureEnv"]
    _MAKE_COUNT[0] += 1
    cls._default_seed = 0xC0FFEE + _MAKE_COUNT[0]
    env = cls(size=9, agents=3)
    T = 160
    ar = np.random.default_rng(31)
    script = ar.choice(7, size=(T, 3), p=[0.12, 0.12, 0.36, 0.12, 0.08, 0.17, 0.03]).astype(np.int8)
    # agent 0: (1, 6) facing down at the key box (1, 7): toggle (box -> key), pickup, turn left (-> facing right/east), forward x3 to
    # (4 - 1, 6)... the door sits at (4, 4): up two cells first
    env.reset(seed=9)
    assert tuple(int(v) for v in env.agents[0].state.pos) == (1, 6) and int(env.agents[0].state.dir) == 1
    plan = [5, 3, 0, 0, 2, 2, 1, 2, 2, 5, 2, 2]      # toggle, pickup, left, left (facing up), fwd, fwd (1,4), right (facing east), fwd, fwd (3,4), toggle door, fwd, fwd
    script[:len(plan), 0] = plan
    record("boxkey_a3", env, "empty", 9, T, None, script=script,
           note="BoxTreasureEnv (tests/custom_envs.py): boxes that hold a key / ball / goal / door / lava; agent 0 opens the key box, "
                "takes the key and unlocks the purple door")


def spec_of_noreset(env, kind):
    return dict(
        env_kind=kind, width=int(env.width), height=int(env.height), num_agents=int(env.num_agents),
        view_size=int(env.agents[0].view_size), see_through_walls=bool(env.agents[0].see_through_walls),
        allow_agent_overlap=bool(env.allow_agent_overlap), joint_reward=bool(env.joint_reward),
        success_termination_mode=str(env.success_termination_mode),
        failure_termination_mode=str(env.failure_termination_mode), max_steps=int(env.max_steps))


if __name__ == "__main__":
    main()
