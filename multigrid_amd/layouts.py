"""Host-side layout generators: the initial (grid, agents) tensors of an episode.

These restate the reference's `_gen_grid` implementations draw for draw, so that -- given numpy generators in
the same state as the reference's -- they produce the identical initial state:

* `empty_layout`                multigrid/envs/empty.py:151-170
* `blockedunlockpickup_layout`  multigrid/envs/blockedunlockpickup.py:142-164 over
                                multigrid/core/roomgrid.py:203-236 (rooms), 238-283 (add_object/place_in_room),
                                285-333 (add_door), 376-404 (place_agent); multigrid/base.py:604-697
                                (place_obj / put_obj / place_agent); multigrid/utils/random.py:9-103

Two generators are involved: `layout_rng` plays the reference's construction-time
generator captured by `RandomMixin` (every `_rand_*` placement draw), `np_random` plays the seeded
`env.np_random` (only `Room.set_door_pos`, roomgrid.py:106; it is the same stream the action order is later
drawn from, so its post-reset state is what goes to the device).

Output layout is the product's (include/mgx.h): grid u8[H,W,3] ([y][x]), agents u8[A,8].
"""
from __future__ import annotations

import numpy as np

from .constants import (AG_CARRY, AG_COLOR, AG_DIR, AG_TERMINATED, AG_X, AG_Y, DIR_TO_VEC, EMPTY_CELL, GOAL_CELL,
                        WALL_CELL, Color, State, Type)


class _Grid:
    """(W,H,3) int state indexed [x, y] like multigrid/core/grid.py:54, walls via wall_rect (grid.py:133-195)."""

    def __init__(self, width: int, height: int):
        self.width, self.height = width, height
        self.state = np.empty((width, height, 3), dtype=np.int64)
        self.state[...] = EMPTY_CELL

    def is_empty(self, x, y) -> bool:           # `grid.get(x, y) is None`
        return self.state[x, y, 0] == Type.empty

    def set(self, x, y, cell):
        self.state[x, y] = EMPTY_CELL if cell is None else cell

    def wall_rect(self, x, y, w, h):
        self.state[x:x + w, y] = WALL_CELL
        self.state[x:x + w, y + h - 1] = WALL_CELL
        self.state[x, y:y + h] = WALL_CELL
        self.state[x + w - 1, y:y + h] = WALL_CELL

    def to_product(self) -> np.ndarray:
        return np.ascontiguousarray(self.state.transpose(1, 0, 2)).astype(np.uint8)


def _fresh_agents(num_agents: int) -> np.ndarray:
    """multigrid/core/agent.py:234-254 + Agent.reset (agent.py:120-133): pos (-1,-1), dir -1, colors cycle."""
    ag = np.zeros((num_agents, 9), dtype=np.int64)
    ag[:, 0] = Type.agent
    ag[:, 1] = [int(c) for c in Color.cycle(num_agents)]
    ag[:, 2] = -1
    ag[:, 3:5] = -1
    ag[:, 6:9] = EMPTY_CELL
    return ag


def pack_agents(ag9: np.ndarray) -> np.ndarray:
    """(A,9) reference rows -> u8[A,8] packed rows."""
    ag9 = np.asarray(ag9)
    assert (ag9[:, 2] >= 0).all() and (ag9[:, 3:5] >= 0).all(), "unplaced agent (multigrid/base.py:283-284)"
    out = np.zeros(ag9.shape[:-1] + (8,), dtype=np.uint8)
    out[..., AG_COLOR] = ag9[..., 1]
    out[..., AG_DIR] = ag9[..., 2]
    out[..., AG_X] = ag9[..., 3]
    out[..., AG_Y] = ag9[..., 4]
    out[..., AG_TERMINATED] = ag9[..., 5]
    out[..., AG_CARRY:AG_CARRY + 3] = ag9[..., 6:9]
    return out


def unpack_agents(ag8: np.ndarray) -> np.ndarray:
    """u8[...,A,8] packed rows -> (...,A,9) int64 reference rows (multigrid/core/agent.py:222-232)."""
    ag8 = np.asarray(ag8)
    out = np.zeros(ag8.shape[:-1] + (9,), dtype=np.int64)
    out[..., 0] = Type.agent
    out[..., 1] = ag8[..., AG_COLOR]
    out[..., 2] = ag8[..., AG_DIR]
    out[..., 3] = ag8[..., AG_X]
    out[..., 4] = ag8[..., AG_Y]
    out[..., 5] = ag8[..., AG_TERMINATED]
    out[..., 6:9] = ag8[..., AG_CARRY:AG_CARRY + 3]
    return out


def grid_to_product(state_whc: np.ndarray) -> np.ndarray:
    """Reference Grid.state (...,W,H,3) -> product grid u8[...,H,W,3]."""
    s = np.asarray(state_whc)
    return np.ascontiguousarray(np.swapaxes(s, -3, -2)).astype(np.uint8)


def grid_from_product(grid_hwc: np.ndarray) -> np.ndarray:
    return np.ascontiguousarray(np.swapaxes(np.asarray(grid_hwc), -3, -2)).astype(np.int64)


# ---- multigrid/base.py:604-697 -------------------------------------------------------------------------
def _place_obj(grid: _Grid, ag: np.ndarray, rng, cell, top=None, size=None, reject_fn=None, max_tries=np.inf):
    top = (0, 0) if top is None else (max(top[0], 0), max(top[1], 0))
    if size is None:
        size = (grid.width, grid.height)
    num_tries = 0
    while True:
        if num_tries > max_tries:
            raise RecursionError("rejection sampling failed in place_obj")
        num_tries += 1
        pos = (int(rng.integers(top[0], min(top[0] + size[0], grid.width))),
               int(rng.integers(top[1], min(top[1] + size[1], grid.height))))
        if not grid.is_empty(*pos):
            continue
        if ((ag[:, 3] == pos[0]) & (ag[:, 4] == pos[1])).any():
            continue
        if reject_fn is not None and reject_fn(ag, pos):
            continue
        break
    grid.set(pos[0], pos[1], cell)
    return pos


def _place_agent(grid: _Grid, ag: np.ndarray, i: int, rng, top=None, size=None, rand_dir=True, max_tries=np.inf):
    ag[i, 3:5] = (-1, -1)
    pos = _place_obj(grid, ag, rng, None, top, size, max_tries=max_tries)
    ag[i, 3:5] = pos
    if rand_dir:
        ag[i, 2] = int(rng.integers(0, 4))
    return pos


def _reject_next_to(ag: np.ndarray, pos) -> bool:
    """multigrid/core/roomgrid.py:45-50"""
    return bool((np.linalg.norm(np.asarray(pos) - ag[:, 3:5], axis=-1) <= 1).any())


# ---- multigrid/envs/empty.py:151-170 --------------------------------------------------------------------
def empty_layout(size: int, num_agents: int, agent_start_pos=(1, 1), agent_start_dir=0, layout_rng=None):
    grid = _Grid(size, size)
    grid.wall_rect(0, 0, size, size)
    grid.set(size - 2, size - 2, GOAL_CELL)
    ag = _fresh_agents(num_agents)
    for i in range(num_agents):
        if agent_start_pos is not None and agent_start_dir is not None:
            ag[i, 3:5] = agent_start_pos
            ag[i, 2] = int(agent_start_dir)
        else:
            _place_agent(grid, ag, i, layout_rng)
    return grid.to_product(), pack_agents(ag)


# ---- multigrid/envs/blockedunlockpickup.py:142-164 ------------------------------------------------------
def blockedunlockpickup_layout(room_size: int, num_agents: int, layout_rng, np_random):
    """Returns (grid u8[H,W,3], agents u8[A,8], target u8[4] = target box (type, color, state, 0))."""
    rs, num_rows, num_cols = room_size, 1, 2
    width, height = (rs - 1) * num_cols + 1, (rs - 1) * num_rows + 1
    grid = _Grid(width, height)
    ag = _fresh_agents(num_agents)
    rooms = {}
    for row in range(num_rows):                                    # roomgrid.py:209-218
        for col in range(num_cols):
            top = (col * (rs - 1), row * (rs - 1))
            rooms[col, row] = (top, (rs, rs))
            grid.wall_rect(*top, rs, rs)
    ag[:, 2] = 0                                                    # roomgrid.py:232-236
    ag[:, 3] = (num_cols // 2) * (rs - 1) + rs // 2
    ag[:, 4] = (num_rows // 2) * (rs - 1) + rs // 2

    def rand_color():                                               # random.py:83-89
        return int(layout_rng.integers(0, len(Color)))

    def place_in_room(col, row, cell):                              # roomgrid.py:238-259
        top, size = rooms[col, row]
        return _place_obj(grid, ag, layout_rng, cell, top, size, reject_fn=_reject_next_to, max_tries=1000)

    # box in the right room (blockedunlockpickup.py:147): add_object(1, 0, kind=box) -> random colour
    box = (int(Type.box), rand_color(), 0)
    place_in_room(1, 0, box)
    # locked door between the rooms (blockedunlockpickup.py:150): colour from layout_rng, row from np_random
    door_color = rand_color()
    (left, top_y), (w, h) = rooms[0, 0]
    right, bottom = left + w - 1, top_y + h - 1
    door_pos = (right, int(np_random.integers(top_y + 1, bottom)))  # roomgrid.py:104-106
    grid.set(*door_pos, (int(Type.door), door_color, int(State.locked)))
    # ball blocking the door (blockedunlockpickup.py:153)
    grid.set(door_pos[0] - 1, door_pos[1], (int(Type.ball), rand_color(), 0))
    # key of the door's colour in the left room (blockedunlockpickup.py:156)
    place_in_room(0, 0, (int(Type.key), door_color, 0))
    # agents in the left room, not facing an object (roomgrid.py:376-404)
    top, size = rooms[0, 0]
    for i in range(num_agents):
        while True:
            _place_agent(grid, ag, i, layout_rng, top, size, True, max_tries=1000)
            fx, fy = ag[i, 3:5] + DIR_TO_VEC[ag[i, 2]]
            if grid.is_empty(fx, fy) or grid.state[fx, fy, 0] == Type.wall:
                break
    target = np.array([box[0], box[1], box[2], 0], dtype=np.uint8)
    return grid.to_product(), pack_agents(ag), target


def check_walled(grid_hwc: np.ndarray):
    """The kernels treat an out-of-bounds front cell as impassable; every shipped env has wall borders
    (SURVEY.md App. A.2), which is what makes that equivalent to the reference.  Checked on import."""
    g = np.asarray(grid_hwc)
    border = np.concatenate([g[..., 0, :, 0].ravel(), g[..., -1, :, 0].ravel(),
                             g[..., :, 0, 0].ravel(), g[..., :, -1, 0].ravel()])
    if not (border == Type.wall).all():
        raise ValueError("grid borders must be walls")
