"""py_oracle.py -- a pure Python / NumPy restatement of ini/multigrid's step / observation path, one env at a time.

TEST INFRASTRUCTURE (only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline legs may import oracle/).  This is the
"reference NumPy path" stand-in SURVEY.md section 8d(i) asks bench.py to time on ONE host core beside the GPU numbers: the
reference itself cannot travel to the GPU box, and its two hot kernels are numba functions that, without numba, run as exactly this
kind of interpreter loop.  It follows the reference's own structure -- per-env Python objects replaced by the two int arrays the
reference keeps underneath them:

    grid_state   (W, H, 3) int   Grid.state             multigrid/core/grid.py:54
    agent_state  (A, 9) int      AgentState rows        multigrid/core/agent.py:222-232
    rng          numpy Generator env.np_random          (the action order: multigrid/base.py:396-399)

Pinned bit for bit by tests/test_oracle_golden.py against fixtures the REAL reference produced (tests/golden/empty*, bup*).  Hook:
BlockedUnlockPickup only (envs/blockedunlockpickup.py:166-175); no box contents (the C oracle, oracle/mgx_oracle.c, has them)."""
from __future__ import annotations

import numpy as np

# multigrid/core/constants.py:34-48, 91-97; multigrid/core/actions.py:5-15
EMPTY, WALL, FLOOR, DOOR, KEY, BALL, BOX, GOAL, LAVA, AGENT = 1, 2, 3, 4, 5, 6, 7, 8, 9, 10
OPEN, CLOSED, LOCKED = 0, 1, 2
LEFT, RIGHT, FORWARD, PICKUP, DROP, TOGGLE, DONE = range(7)
DIR_TO_VEC = ((1, 0), (0, 1), (-1, 0), (0, -1))              # constants.py:21-30
WALL_ENCODING = (2, 5, 0)                                     # utils/obs.py:14
UNSEEN_ENCODING = (0, 0, 0)                                   # utils/obs.py:15
EMPTY_ENCODING = (1, 0, 0)                                    # world_object.py:131-137


def see_behind(cell) -> bool:
    """utils/obs.py:46-63"""
    if cell[0] == WALL:
        return False
    if cell[0] == DOOR and cell[2] != OPEN:
        return False
    return True


def get_view_exts(agent_state, v):
    """utils/obs.py:275-316: top-left corner of every agent's view"""
    out = []
    for s in agent_state:
        d, x, y = int(s[2]), int(s[3]), int(s[4])
        if d == 0:
            out.append((x, y - v // 2))
        elif d == 1:
            out.append((x - v // 2, y))
        elif d == 2:
            out.append((x - v + 1, y - v // 2))
        else:
            out.append((x - v // 2, y - v + 1))
    return out


def gen_obs_grid(grid_state, agent_state, v):
    """utils/obs.py:130-209: agent overlay, view window, rotation to the agent's facing, walls outside, the carried object"""
    W, H = grid_state.shape[:2]
    A = len(agent_state)
    grid_encoding = grid_state
    if A > 1:                                                 # obs.py:163-173
        grid_encoding = grid_state.copy()
        for s in agent_state:
            if not s[5]:
                grid_encoding[s[3], s[4]] = (s[0], s[1], s[2])
    top_left = get_view_exts(agent_state, v)
    obs = np.empty((A, v, v, 3), dtype=np.int64)
    for a in range(A):
        s = agent_state[a]
        rot = (int(s[2]) + 1) % 4
        tx, ty = top_left[a]
        for i in range(v):
            for j in range(v):
                x, y = tx + i, ty + j
                if rot == 0:
                    ir, jr = i, j
                elif rot == 1:
                    ir, jr = j, v - i - 1
                elif rot == 2:
                    ir, jr = v - i - 1, v - j - 1
                else:
                    ir, jr = v - j - 1, i
                obs[a, ir, jr] = grid_encoding[x, y] if (0 <= x < W and 0 <= y < H) else WALL_ENCODING   # obs.py:199-202
        obs[a, v // 2, v - 1] = s[6:9]                        # obs.py:207
    return obs


def get_vis_mask(obs_grid):
    """utils/obs.py:235-273: the two sweeps per depth row, verbatim"""
    A, v = obs_grid.shape[:2]
    vis = np.zeros((A, v, v), dtype=bool)
    for a in range(A):
        sb = [[see_behind(obs_grid[a, i, j]) for j in range(v)] for i in range(v)]
        m = vis[a]
        m[v // 2, v - 1] = True
        for j in range(v - 1, -1, -1):
            for i in range(0, v - 1):
                if m[i, j] and sb[i][j]:
                    m[i + 1, j] = True
                    if j > 0:
                        m[i + 1, j - 1] = True
                        m[i, j - 1] = True
            for i in range(v - 1, 0, -1):
                if m[i, j] and sb[i][j]:
                    m[i - 1, j] = True
                    if j > 0:
                        m[i - 1, j - 1] = True
                        m[i, j - 1] = True
    return vis


def gen_obs(grid_state, agent_state, v, see_through_walls=False):
    """utils/obs.py:65-102 gen_obs_grid_encoding -> (A, v, v, 3)"""
    obs = gen_obs_grid(grid_state, agent_state, v)
    if not see_through_walls:
        vis = get_vis_mask(obs)
        obs[~vis] = UNSEEN_ENCODING
    return obs


def _reward(step_count, max_steps):
    return 1 - 0.9 * (step_count / max_steps)                 # base.py:598-602


def _on_success(spec, agent_state, i, step_count, rewards):
    """base.py:478-507"""
    if spec["success_termination_mode"] == "any":
        agent_state[:, 5] = 1
    else:
        agent_state[i, 5] = 1
    r = _reward(step_count, spec["max_steps"])
    if spec["joint_reward"]:
        rewards[:] = r
    else:
        rewards[i] = r


def _on_failure(spec, agent_state, i):
    """base.py:509-532"""
    if spec["failure_termination_mode"] == "any":
        agent_state[:, 5] = 1
    else:
        agent_state[i, 5] = 1


def handle_actions(spec, grid_state, agent_state, rng, step_count, actions):
    """base.py:378-476.  actions: (A,) ints, -1 = the agent's key is absent from the dict.  Returns rewards (A,) float64."""
    A = len(agent_state)
    rewards = np.zeros(A, dtype=np.float64)
    order = [0] if A == 1 else rng.random(size=A).argsort()   # base.py:396-399
    for i in order:
        a = int(actions[i])
        s = agent_state[i]
        if a < 0 or s[5]:                                     # base.py:403-409
            continue
        dx, dy = DIR_TO_VEC[int(s[2])]
        fx, fy = int(s[3]) + dx, int(s[4]) + dy
        cell = grid_state[fx, fy]
        if a == LEFT:
            s[2] = (s[2] - 1) % 4
        elif a == RIGHT:
            s[2] = (s[2] + 1) % 4
        elif a == FORWARD:                                    # base.py:420-436
            t = cell[0]
            if t in (EMPTY, FLOOR, GOAL, LAVA) or (t == DOOR and cell[2] == OPEN):
                if not spec["allow_agent_overlap"] and ((agent_state[:, 3] == fx) & (agent_state[:, 4] == fy)).any():
                    continue
                s[3], s[4] = fx, fy
                if t == GOAL:
                    _on_success(spec, agent_state, i, step_count, rewards)
                if t == LAVA:
                    _on_failure(spec, agent_state, i)
        elif a == PICKUP:                                     # base.py:439-446
            if cell[0] in (KEY, BALL, BOX) and s[6] == EMPTY:
                s[6:9] = cell
                grid_state[fx, fy] = EMPTY_ENCODING
        elif a == DROP:                                       # base.py:449-459
            if s[6] != EMPTY and cell[0] == EMPTY and not ((agent_state[:, 3] == fx) & (agent_state[:, 4] == fy)).any():
                grid_state[fx, fy] = s[6:9]
                s[6:9] = EMPTY_ENCODING
        elif a == TOGGLE:                                     # base.py:462-467; world_object.py:458-474, 599-605
            if cell[0] == DOOR:
                if cell[2] == LOCKED:
                    if s[6] == KEY and s[7] == cell[1]:
                        cell[2] = OPEN
                elif cell[2] == OPEN:
                    cell[2] = CLOSED
                else:
                    cell[2] = OPEN
            elif cell[0] == BOX:
                grid_state[fx, fy] = EMPTY_ENCODING           # (contains is None)
        elif a == DONE:
            pass
        else:
            raise ValueError(f"Unknown action: {a}")          # base.py:473-474
    return rewards


def step(spec, grid_state, agent_state, rng, step_count, actions, target=None):
    """base.py:303-346 (+ envs/blockedunlockpickup.py:166-175).  The arrays are updated in place.
    Returns (obs (A,v,v,3), direction (A,), rewards (A,), terminated (A,), truncated, step_count)."""
    step_count += 1                                           # base.py:333
    rewards = handle_actions(spec, grid_state, agent_state, rng, step_count, actions)
    obs = gen_obs(grid_state, agent_state, spec["view_size"], spec["see_through_walls"])
    terminated = agent_state[:, 5].astype(bool).copy()        # base.py:338
    truncated = step_count >= spec["max_steps"]
    direction = agent_state[:, 2].copy()
    if spec.get("env_kind") == "blockedunlockpickup":
        for i in range(len(agent_state)):
            if agent_state[i, 6] == target[0] and agent_state[i, 7] == target[1]:       # `carrying == self.obj`
                _on_success(spec, agent_state, i, step_count, rewards)
                if spec["success_termination_mode"] == "any":
                    terminated[:] = True
                else:
                    terminated[i] = True
    return obs, direction, rewards, terminated, truncated, step_count
