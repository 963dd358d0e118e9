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
    """Product grid u8[...,H,W,3] -> the reference's Grid.state (...,W,H,3)."""
    return np.ascontiguousarray(np.swapaxes(np.asarray(grid_hwc), -3, -2)).astype(np.int64)


def pack_cells(grid_hwc: np.ndarray) -> np.ndarray:
    """(type, color, state) bytes u8[...,3] -> the device's packed cells u16[...] (include/mgx.h MgxCell):
    [3:0] type | [10:8] color | [13:12] state | [15] opaque, opaque = not see_behind (multigrid/utils/obs.py:46-63: a wall,
    or a door that is not open).  A BOX's state byte may carry what the box holds (include/mgx.h "BOX CONTENTS": state | kind << 2
    | colour << 5): kind -> bits [6:4], colour -> bits 7, 11, 14.  Values the 16 bits cannot hold (the reference has none) are
    refused."""
    g = np.asarray(grid_hwc)
    if g.shape[-1] != 3:
        raise ValueError("pack_cells expects (type, color, state) triples in the last axis")
    t, c, sb = (g[..., k].astype(np.uint16) for k in range(3))
    s, kind, ccol = sb & 3, (sb >> 2) & 7, (sb >> 5) & 7
    if (t > 15).any() or (c > 7).any():
        raise ValueError("cell value outside the packed format (type <= 15, color <= 7, state <= 3)")
    has = (sb >> 2) != 0
    if (has & ((t != Type.box) | (kind == 0) | (ccol > 5))).any():
        raise ValueError("cell value outside the packed format: a state byte above 3 is a BOX's content (kind 1..7, colour <= 5)")
    opaque = (t == Type.wall) | ((t == Type.door) & (s != State.open))
    content = (kind << 4) | ((ccol & 1) << 7) | (((ccol >> 1) & 1) << 11) | (((ccol >> 2) & 1) << 14)
    return (t | (c << 8) | (s << 12) | (opaque.astype(np.uint16) << 15) | content).astype(np.uint16)


def unpack_cells(cells: np.ndarray) -> np.ndarray:
    """Packed cells u16[...] (or their int16 view) -> (type, color, state) bytes u8[...,3]."""
    p = np.asarray(cells)
    p = p.view(np.uint16) if p.dtype == np.int16 else p.astype(np.uint16)
    content = ((p >> 4) & 0xf) | (((p >> 11) & 1) << 4) | (((p >> 14) & 1) << 5)          # a box's content: kind | colour << 3
    return np.stack((p & 0xf, (p >> 8) & 0x7, ((p >> 12) & 0x3) | (content << 2)), axis=-1).astype(np.uint8)


def pack_cells8(grid_hwc: np.ndarray) -> np.ndarray:
    """(type, color, state) bytes u8[...,3] -> COMPACT cells u8[...] (include/mgx.h MgxCell8): tcode [3:0] | color [6:4] | opaque [7],
    tcode = type for state 0, 11 / 12 = closed / locked door, 13..15 = agent overlay facing 1..3.  Refuses what one byte cannot
    hold: a state on anything but a door or an agent overlay."""
    g = np.asarray(grid_hwc)
    if g.shape[-1] != 3:
        raise ValueError("pack_cells8 expects (type, color, state) triples in the last axis")
    t, c, s = (g[..., k].astype(np.uint16) for k in range(3))
    if (t > int(Type.agent)).any() or (c > 7).any() or (s > 3).any():
        raise ValueError("cell value outside the compact format's range (type <= 10, color <= 7, state <= 3; a box's content "
                         "needs the 16-bit cells)")
    if ((s != 0) & (t != int(Type.door)) & (t != int(Type.agent))).any():
        raise ValueError("the compact cell format holds a state only on doors and agent overlays")
    tc = np.where(s == 0, t, np.where(t == int(Type.door), 10 + s, 12 + s))
    opaque = (t == int(Type.wall)) | ((t == int(Type.door)) & (s != 0))
    return (tc | (c << 4) | (opaque.astype(np.uint16) << 7)).astype(np.uint8)


def unpack_cells8(cells: np.ndarray) -> np.ndarray:
    """COMPACT cells u8[...] -> (type, color, state) bytes u8[...,3]."""
    p = np.asarray(cells).astype(np.uint8).astype(np.int32)
    tc = p & 15
    door = (tc == 11) | (tc == 12)
    t = np.where(tc <= 10, tc, np.where(door, int(Type.door), int(Type.agent)))
    s = np.where(tc <= 10, 0, np.where(door, tc - 10, tc - 12))
    return np.stack((t, (p >> 4) & 7, s), axis=-1).astype(np.uint8)


def pack_cells_for(spec, grid_hwc: np.ndarray) -> np.ndarray:
    """The device's cell tensor content for `spec` (its `cell_bytes`): i16 / u8 bit patterns."""
    if spec.cell_bytes == 3:                                  # byte grids: the triples themselves (checked like the packed forms)
        pack_cells(grid_hwc)
        return np.ascontiguousarray(grid_hwc, dtype=np.uint8)
    return pack_cells8(grid_hwc) if spec.cell_bytes == 1 else pack_cells(grid_hwc).view(np.int16)


def unpack_cells_for(spec, cells: np.ndarray) -> np.ndarray:
    if spec.cell_bytes == 3:
        return np.asarray(cells, dtype=np.uint8)
    return unpack_cells8(cells) if spec.cell_bytes == 1 else unpack_cells(cells)


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


def roomgrid_blank(room_size: int, num_rows: int, num_cols: int) -> np.ndarray:
    """The walls of a RoomGrid before any object is placed (multigrid/core/roomgrid.py:203-218): u8[H,W,3].  The
    template the on-device generator starts an episode from (mgx_reset_generate)."""
    rs = room_size
    grid = _Grid((rs - 1) * num_cols + 1, (rs - 1) * num_rows + 1)
    for row in range(num_rows):
        for col in range(num_cols):
            grid.wall_rect(col * (rs - 1), row * (rs - 1), rs, rs)
    return grid.to_product()


def empty_blank(size: int) -> np.ndarray:
    """EmptyEnv's grid before the agents are placed (multigrid/envs/empty.py:156-162): border walls + the goal."""
    grid = _Grid(size, size)
    grid.wall_rect(0, 0, size, size)
    grid.set(size - 2, size - 2, GOAL_CELL)
    return grid.to_product()


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


# ---- multigrid/core/roomgrid.py:53-495 (Room, RoomGrid) restated over a _Grid -------------------------------------
class _Room:
    def __init__(self, top, size):
        self.top, self.size = top, size
        self.doors = {d: None for d in range(4)}          # None | True (wall removed) | door cell tuple
        self.door_pos = {d: None for d in range(4)}
        self.neighbors = {d: None for d in range(4)}
        self.objs = []

    @property
    def locked(self) -> bool:                              # roomgrid.py:84-88
        return any(isinstance(door, tuple) and door[2] == State.locked for door in self.doors.values())

    def set_door_pos(self, d, random=None):                # roomgrid.py:90-128
        left, top = self.top
        right, bottom = left + self.size[0] - 1, top + self.size[1] - 1
        if d == 0:
            pos = (right, int(random.integers(top + 1, bottom))) if random is not None else (right, (top + bottom) // 2)
        elif d == 1:
            pos = (int(random.integers(left + 1, right)), bottom) if random is not None else ((left + right) // 2, bottom)
        elif d == 2:
            pos = (left, int(random.integers(top + 1, bottom))) if random is not None else (left, (top + bottom) // 2)
        else:
            pos = (int(random.integers(left + 1, right)), top) if random is not None else ((left + right) // 2, top)
        self.door_pos[d] = pos
        return pos


class _RoomGrid:
    """RoomGrid._gen_grid and helpers (roomgrid.py:203-463); `rng` = construction-time generator, `np_random` = seeded."""

    def __init__(self, room_size, num_rows, num_cols, num_agents, rng, np_random):
        self.rs, self.num_rows, self.num_cols = room_size, num_rows, num_cols
        self.rng, self.np_random = rng, np_random
        self.width, self.height = (room_size - 1) * num_cols + 1, (room_size - 1) * num_rows + 1
        self.grid = _Grid(self.width, self.height)
        self.ag = _fresh_agents(num_agents)
        self.room_grid = [[None] * num_cols for _ in range(num_rows)]
        for row in range(num_rows):                                    # roomgrid.py:209-218
            for col in range(num_cols):
                room = _Room((col * (room_size - 1), row * (room_size - 1)), (room_size, room_size))
                self.room_grid[row][col] = room
                self.grid.wall_rect(*room.top, *room.size)
        for row in range(num_rows):                                    # roomgrid.py:220-231
            for col in range(num_cols):
                room = self.room_grid[row][col]
                if col < num_cols - 1:
                    room.neighbors[0] = self.room_grid[row][col + 1]
                if row < num_rows - 1:
                    room.neighbors[1] = self.room_grid[row + 1][col]
                if col > 0:
                    room.neighbors[2] = self.room_grid[row][col - 1]
                if row > 0:
                    room.neighbors[3] = self.room_grid[row - 1][col]
        self.ag[:, 2] = 0                                              # roomgrid.py:232-236
        self.ag[:, 3] = (num_cols // 2) * (room_size - 1) + room_size // 2
        self.ag[:, 4] = (num_rows // 2) * (room_size - 1) + room_size // 2

    def get_room(self, col, row):
        return self.room_grid[row][col]

    def rand_int(self, lo, hi):
        return int(self.rng.integers(lo, hi))

    def rand_color(self):
        return self.rand_int(0, len(Color))

    def place_in_room(self, col, row, cell):                           # roomgrid.py:238-259
        room = self.get_room(col, row)
        pos = _place_obj(self.grid, self.ag, self.rng, cell, room.top, room.size, reject_fn=_reject_next_to, max_tries=1000)
        room.objs.append(cell)
        return pos

    def add_object(self, col, row, kind=None, color=None):             # roomgrid.py:261-283
        kind = kind if kind is not None else [Type.key, Type.ball, Type.box][self.rand_int(0, 3)]
        color = color if color is not None else self.rand_color()
        cell = (int(kind), int(color), 0)
        return cell, self.place_in_room(col, row, cell)

    def add_door(self, col, row, d, color=None, locked=None, rand_pos=True):    # roomgrid.py:285-331
        room = self.get_room(col, row)
        assert room.neighbors[d] is not None, "no neighbor in this direction"
        assert room.doors[d] is None, "door already exists"
        color = color if color is not None else self.rand_color()
        locked = locked if locked is not None else (self.rand_int(0, 2) == 0)
        door = (int(Type.door), int(color), int(State.locked if locked else State.closed))
        pos = room.set_door_pos(d, random=self.np_random if rand_pos else None)
        self.grid.set(*pos, door)
        room.doors[d] = door
        room.neighbors[d].doors[(d + 2) % 4] = door
        return door, pos

    def remove_wall(self, col, row, d):                                 # roomgrid.py:333-374
        room = self.get_room(col, row)
        assert room.doors[d] is None and room.neighbors[d]
        tx, ty = room.top
        w, h = room.size
        for i in range(1, (h if d in (0, 2) else w) - 1):
            x, y = {0: (tx + w - 1, ty + i), 1: (tx + i, ty + h - 1), 2: (tx, ty + i), 3: (tx + i, ty)}[d]
            self.grid.set(x, y, None)
        room.doors[d] = True
        room.neighbors[d].doors[(d + 2) % 4] = True

    def place_agent(self, i, col=None, row=None, rand_dir=True):        # roomgrid.py:376-404
        col = col if col is not None else self.rand_int(0, self.num_cols)
        row = row if row is not None else self.rand_int(0, self.num_rows)
        room = self.get_room(col, row)
        while True:
            _place_agent(self.grid, self.ag, i, self.rng, room.top, room.size, rand_dir, max_tries=1000)
            fx, fy = self.ag[i, 3:5] + DIR_TO_VEC[self.ag[i, 2]]
            if self.grid.is_empty(fx, fy) or self.grid.state[fx, fy, 0] == Type.wall:
                break

    def connect_all(self, door_colors=None, max_itrs=5000):             # roomgrid.py:406-452
        door_colors = list(range(len(Color))) if door_colors is None else door_colors
        start = self.get_room(0, 0)
        for _ in range(max_itrs):
            seen, queue = set(), [start]                                # bfs, roomgrid.py:20-43
            while queue:
                node = queue.pop(0)
                if id(node) not in seen:
                    seen.add(id(node))
                    queue.extend(node.neighbors[d] for d in range(4) if node.doors[d] is not None)
            if len(seen) == self.num_rows * self.num_cols:
                return
            col, row, d = self.rand_int(0, self.num_cols), self.rand_int(0, self.num_rows), self.rand_int(0, 4)
            room = self.get_room(col, row)
            if not room.neighbors[d] or room.doors[d]:
                continue
            if room.locked or room.neighbors[d].locked:
                continue
            color = door_colors[self.rand_int(0, len(door_colors))]
            self.add_door(col, row, d, color=color, locked=False)
        raise RecursionError("connect_all() failed")

    def result(self):
        return self.grid.to_product(), pack_agents(self.ag)


def rules_aux(rules) -> np.ndarray:
    """The hook table of an `env_kind="rules"` env (include/mgx.h: MGX_KIND_RULES) as its aux u8[16]: a user-defined env's
    `step` post-hook, declared.  `rules`: up to 3 tuples, evaluated in order after the base step --
        ("carries", type, color, "success" | "failure")                  `if agent.state.carrying == self.obj: self.on_...(agent)`
        ("toggles_at", x, y, "success" | "failure"[, "open" | "shut"])   `if action == toggle and fwd_obj == self.door [and (not)
                                                                          self.door.is_open]: self.on_...(agent)`
    BlockedUnlockPickup's hook (envs/blockedunlockpickup.py:166-175) is `[("carries", Type.box, color, "success")]`."""
    if len(rules) > 3:
        raise ValueError("at most 3 rules fit the env's 16 bytes of hook state")
    aux = np.zeros(16, dtype=np.uint8)
    aux[0] = len(rules)
    for k, r in enumerate(rules):
        op = {"carries": 1, "toggles_at": 2}[r[0]]
        cond = {"always": 0, "open": 1, "shut": 2}[r[4]] if len(r) > 4 else 0
        aux[1 + 5 * k:6 + 5 * k] = (op, int(r[1]), int(r[2]), {"success": 1, "failure": 2}[r[3]], cond)
    return aux


def make_aux(kind: str, grid_hwc: np.ndarray, target=None) -> np.ndarray:
    """The 16-byte hook state of include/mgx.h for a freshly generated layout."""
    aux = np.zeros(16, dtype=np.uint8)
    g = np.asarray(grid_hwc)
    if kind == "blockedunlockpickup":
        aux[:3] = target[:3]
    elif kind == "redbluedoors":
        (by, bx), = np.argwhere((g[..., 0] == Type.door) & (g[..., 1] == Color.blue))
        (ry, rx), = np.argwhere((g[..., 0] == Type.door) & (g[..., 1] == Color.red))
        aux[:4] = (bx, by, rx, ry)
    elif kind == "lockedhallway":
        doors = sorted((int(x), int(y)) for y, x in np.argwhere(g[..., 0] == Type.door))
        if len(doors) <= 6:                                      # explicit positions (include/mgx.h)
            aux[0] = len(doors)
            for k, (x, y) in enumerate(doors):
                aux[2 + 2 * k], aux[3 + 2 * k] = x, y
        else:                                                    # geometric format: doors sit mid-wall, two per row of rooms
            rs = min(x for x, _ in doors) + 1
            want = sorted(((rs - 1) * (1 + side), row * (rs - 1) + (rs - 1) // 2) for row in range(len(doors) // 2) for side in (0, 1))
            if doors != want or len(doors) > 16:
                raise ValueError("LockedHallway with more than 6 doors: they must sit mid-wall, two per row (at most 16)")
            aux[0] = 0x80 | len(doors)
            aux[3] = rs
            # len(self.rooms): the reference keys its rooms by door colour (locked_hallway.py:166-176), so repeated colours
            # count once -- and that count is what ends the episode (locked_hallway.py:222-225)
            aux[4] = len({int(g[y, x, 1]) for x, y in doors})
    return aux


# ---- multigrid/envs/playground.py:122-137 -----------------------------------------------------------------------------
def playground_layout(room_size, num_rows, num_cols, num_agents, layout_rng, np_random):
    rg = _RoomGrid(room_size, num_rows, num_cols, num_agents, layout_rng, np_random)
    rg.connect_all()
    for _ in range(12):
        col, row = rg.rand_int(0, num_cols), rg.rand_int(0, num_rows)
        rg.add_object(col, row)
    for i in range(num_agents):
        rg.place_agent(i)
    return rg.result()


def lockedhallway_blank(num_rooms: int, room_size: int) -> np.ndarray:
    """LockedHallway before doors, keys and agents are placed (multigrid/envs/locked_hallway.py:155-166): the walls of 3 columns
    of rooms with the hallway's inner walls removed -- the template of the on-device generator (MGX_GEN_LOCKEDHALLWAY)."""
    rg = _RoomGrid(room_size, num_rooms // 2, 3, 1, None, None)
    for row in range(num_rooms // 2 - 1):
        rg.remove_wall(1, row, 1)
    return rg.grid.to_product()


def redbluedoors_blank(size: int) -> np.ndarray:
    """RedBlueDoors before agents and doors are placed (multigrid/envs/redbluedoors.py:144-153): the outer walls and the walls
    of the middle room, u8[H,W,3] -- the template of the on-device generator (mgx_reset_generate, MGX_GEN_REDBLUEDOORS)."""
    width, height = 2 * size, size
    grid = _Grid(width, height)
    grid.wall_rect(0, 0, width, height)
    grid.wall_rect(width // 4, 0, width // 2, height)
    return grid.to_product()


# ---- multigrid/envs/redbluedoors.py:142-168 -----------------------------------------------------------------------------
def redbluedoors_layout(size, num_agents, layout_rng):
    width, height = 2 * size, size
    grid = _Grid(width, height)
    ag = _fresh_agents(num_agents)
    room_top, room_size = (width // 4, 0), (width // 2, height)
    grid.wall_rect(0, 0, width, height)
    grid.wall_rect(*room_top, *room_size)
    for i in range(num_agents):
        _place_agent(grid, ag, i, layout_rng, top=room_top, size=room_size)
    grid.set(room_top[0], int(layout_rng.integers(1, height - 1)), (int(Type.door), int(Color.red), int(State.closed)))
    grid.set(room_top[0] + room_size[0] - 1, int(layout_rng.integers(1, height - 1)),
             (int(Type.door), int(Color.blue), int(State.closed)))
    return grid.to_product(), pack_agents(ag)


# ---- multigrid/envs/locked_hallway.py:155-201 ---------------------------------------------------------------------------
def lockedhallway_layout(num_rooms, room_size, max_hallway_keys, max_keys_per_room, num_agents, layout_rng, np_random):
    from math import ceil
    num_rows = num_rooms // 2
    rg = _RoomGrid(room_size, num_rows, 3, num_agents, layout_rng, np_random)
    LEFT, HALLWAY, RIGHT = range(3)
    color_sequence = list(range(len(Color))) * ceil(num_rooms / len(Color))
    layout_rng.shuffle(color_sequence)                                   # _rand_perm, random.py:75-83
    color_sequence = color_sequence[:num_rooms]
    for row in range(num_rows - 1):
        rg.remove_wall(HALLWAY, row, 1)
    rooms = {}
    door_colors = list(color_sequence)
    layout_rng.shuffle(door_colors)
    for row in range(num_rows):
        for col, d in ((LEFT, 0), (RIGHT, 2)):
            color = door_colors.pop()
            rooms[color] = rg.get_room(col, row)
            rg.add_door(col, row, d, color=color, locked=True, rand_pos=False)
    num_hallway_keys = rg.rand_int(1, max_hallway_keys + 1)
    hallway_top = rg.get_room(HALLWAY, 0).top
    hallway_size = (rg.get_room(HALLWAY, 0).size[0], rg.height)
    for key_color in color_sequence[:num_hallway_keys]:
        _place_obj(rg.grid, rg.ag, layout_rng, (int(Type.key), key_color, 0), hallway_top, hallway_size)
    key_index = num_hallway_keys
    while key_index < len(color_sequence):
        room = rooms[color_sequence[key_index - 1]]
        num_room_keys = rg.rand_int(1, max_keys_per_room + 1)
        for key_color in color_sequence[key_index:key_index + num_room_keys]:
            _place_obj(rg.grid, rg.ag, layout_rng, (int(Type.key), key_color, 0), room.top, room.size)
            key_index += 1
    for i in range(num_agents):
        _place_agent(rg.grid, rg.ag, i, layout_rng, top=hallway_top, size=hallway_size)
    return rg.result()


def check_walled(grid_hwc: np.ndarray):
    """The precondition of include/mgx.h: the outer ring of every grid is the reference's WALL = (wall, grey, 0)
    (multigrid/utils/obs.py:14) -- every shipped env starts from Grid.wall_rect(0, 0, W, H).  The kernels
    rely on it twice: an out-of-bounds front cell is never tested, and a view cell outside the grid is read from the ring cell
    next to it (the clamped gather).  Checked on import."""
    g = np.asarray(grid_hwc)
    ring = np.concatenate([g[..., 0, :, :].reshape(-1, 3), g[..., -1, :, :].reshape(-1, 3),
                           g[..., :, 0, :].reshape(-1, 3), g[..., :, -1, :].reshape(-1, 3)])
    if not (ring[:, 0] == Type.wall).all():
        raise ValueError("grid borders must be walls")
    if not ((ring[:, 1] == Color.grey) & (ring[:, 2] == 0)).all():
        raise ValueError("grid borders must be the reference's WALL cells (wall, grey, 0): multigrid/utils/obs.py:14")
