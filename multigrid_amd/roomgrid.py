"""`Room` and `RoomGrid`: the reference's base class for environments made of rooms (multigrid/core/roomgrid.py:53-495), as a
public class over this package's `MultiGridEnv` -- three of the reference's five env classes derive from it
(envs/blockedunlockpickup.py, locked_hallway.py, playground.py) and so do user envs written like them:

    from multigrid_amd.core.roomgrid import RoomGrid            # `from multigrid.core.roomgrid import RoomGrid`

    class TwoRooms(RoomGrid):
        def __init__(self, **kwargs):
            super().__init__(room_size=6, num_rows=1, num_cols=2, max_steps=200, **kwargs)
        def _gen_grid(self, width, height):
            super()._gen_grid(width, height)
            self.box, _ = self.add_object(1, 0, kind=Type.box)
            self.door, _ = self.add_door(0, 0, Direction.right, locked=True)
            self.add_object(0, 0, Type.key, self.door.color)
            for agent in self.agents:
                self.place_agent(agent, 0, 0)

Everything here runs on the host inside `reset` (it builds the episode's first grid; `MultiGridEnv._gen_layout` then packs and
uploads it), through the same placement helpers as any user `_gen_grid` -- `place_obj`'s two draws per attempt, `_rand_elem`'s one --
so a subclass draws from the construction-time generator and from `np_random` (door positions, roomgrid.py:324) in the reference's
order and reproduces its layouts bit for bit (tests/test_custom_envs.py: a RoomGrid env recorded over the real reference).  The
built-in env classes keep their own array-level generators (`layouts.py`: the same draws without objects), which is what the layout
pools and the device generators are pinned against.

Kept as the reference has them, because a drop-in that "fixes" them would produce other layouts or other errors:
* `add_door(col, row)` without a direction looks up `room.neighbors[None]` first (roomgrid.py:316) -> KeyError;
* `add_distractors` keeps the (type, colour) keys it has seen in a set and `.append`s to it (roomgrid.py:478, 493): the FIRST
  distractor is placed, then AttributeError -- unless `num_distractors` is 0.
"""
from __future__ import annotations

from collections import deque
from typing import Callable, Iterable, TypeVar

import numpy as np

from .constants import Color, Direction, Type
from .env import MultiGridEnv
from .world import Door, Grid, WorldObj

T = TypeVar("T")


def bfs(start_node: T, neighbor_fn: Callable[[T], Iterable[T]]) -> set:
    """Nodes reachable from `start_node` (roomgrid.py:20-43)."""
    seen: set = set()
    todo = deque((start_node,))
    while todo:
        node = todo.popleft()
        if node in seen:
            continue
        seen.add(node)
        todo.extend(neighbor_fn(node))
    return seen


def reject_next_to(env: MultiGridEnv, pos) -> bool:
    """`place_obj` filter: no object on or right beside an agent's starting cell (roomgrid.py:45-50: Euclidean distance <= 1)."""
    gap = np.asarray(pos) - np.asarray(env.agent_states.pos).reshape(-1, 2)
    return bool((np.hypot(gap[:, 0], gap[:, 1]) <= 1).any())


def _opposite(dir) -> int:
    return (int(dir) + 2) % 4


class Room:
    """A rectangle of the grid, walls included, that knows its neighbours and the doors in its four walls (roomgrid.py:53-136).
    `doors[d]` is None (solid wall), a `Door`, or True (the wall was removed)."""

    def __init__(self, top: tuple[int, int], size: tuple[int, int]):
        self.top, self.size = top, size
        self.doors: dict = {d: None for d in Direction}
        self.door_pos: dict = {d: None for d in Direction}
        self.neighbors: dict = {d: None for d in Direction}
        self.objs: list = []

    @property
    def locked(self) -> bool:
        """Is this room behind a locked door?"""
        return any(door and door.is_locked for door in self.doors.values())

    def _wall_span(self, dir):
        """The wall on side `dir`: (fixed coordinate, first and last cell of the wall's run, is the run vertical?)."""
        x0, y0 = self.top
        x1, y1 = x0 + self.size[0] - 1, y0 + self.size[1] - 1
        return {Direction.right: (x1, y0, y1, True), Direction.down: (y1, x0, x1, False),
                Direction.left: (x0, y0, y1, True), Direction.up: (y0, x0, x1, False)}[Direction(dir)]

    def set_door_pos(self, dir, random: np.random.Generator | None = None) -> tuple[int, int]:
        """Where the door of wall `dir` goes: the middle of the wall, or -- with a generator -- one `integers` draw over the cells
        between its corners (roomgrid.py:90-128)."""
        fixed, lo, hi, vertical = self._wall_span(dir)
        along = random.integers(lo + 1, hi) if random else (lo + hi) // 2
        self.door_pos[dir] = (fixed, along) if vertical else (along, fixed)
        return self.door_pos[dir]

    def pos_inside(self, x: int, y: int) -> bool:
        return self.top[0] <= x < self.top[0] + self.size[0] and self.top[1] <= y < self.top[1] + self.size[1]


class RoomGrid(MultiGridEnv):
    """`num_rows` x `num_cols` square rooms of side `room_size` sharing their walls (roomgrid.py:139-495)."""

    def __init__(self, room_size: int = 7, num_rows: int = 3, num_cols: int = 3, **kwargs):
        assert room_size >= 3
        assert num_rows > 0
        assert num_cols > 0
        self.room_size, self.num_rows, self.num_cols = room_size, num_rows, num_cols
        span = room_size - 1
        super().__init__(width=span * num_cols + 1, height=span * num_rows + 1, **kwargs)

    # ------------------------------------------------------------------------------------------ rooms
    def get_room(self, col: int, row: int) -> Room:
        assert 0 <= col < self.num_cols
        assert 0 <= row < self.num_rows
        return self.room_grid[row][col]

    def room_from_pos(self, x: int, y: int) -> Room:
        span = self.room_size - 1
        return self.get_room(x // span, y // span)

    def _gen_grid(self, width, height):
        """Walls of every room, the rooms' neighbour links, all agents in the middle of the middle room facing right
        (roomgrid.py:203-236)."""
        self.grid = Grid(width, height)
        span, side = self.room_size - 1, self.room_size
        self.room_grid = [[Room((col * span, row * span), (side, side)) for col in range(self.num_cols)]
                          for row in range(self.num_rows)]
        for row, rooms in enumerate(self.room_grid):
            for col, room in enumerate(rooms):
                self.grid.wall_rect(*room.top, *room.size)
                for d, (dc, dr) in ((Direction.right, (1, 0)), (Direction.down, (0, 1)), (Direction.left, (-1, 0)),
                                    (Direction.up, (0, -1))):
                    if 0 <= col + dc < self.num_cols and 0 <= row + dr < self.num_rows:
                        room.neighbors[d] = self.room_grid[row + dr][col + dc]
        self.agent_states.dir = Direction.right
        self.agent_states.pos = ((self.num_cols // 2) * span + side // 2, (self.num_rows // 2) * span + side // 2)

    # ------------------------------------------------------------------------------------------ objects
    def place_in_room(self, col: int, row: int, obj: WorldObj):
        """`obj` on a free cell of the room, not beside an agent's start (roomgrid.py:238-259).  Returns (obj, pos)."""
        room = self.get_room(col, row)
        pos = self.place_obj(obj, room.top, room.size, reject_fn=reject_next_to, max_tries=1000)
        room.objs.append(obj)
        return obj, pos

    def add_object(self, col: int, row: int, kind=None, color=None):
        """A new key / ball / box (one draw each for a kind and a colour that are not given) in the room (roomgrid.py:261-283)."""
        kind = kind or self._rand_elem([Type.key, Type.ball, Type.box])         # (enum members are truthy, as the reference's
        color = color or self._rand_color()                                     # strings: Color.red, index 0, is kept)
        return self.place_in_room(col, row, WorldObj(type=kind, color=color))

    def add_door(self, col: int, row: int, dir=None, color=None, locked=None, rand_pos: bool = True):
        """A door in wall `dir` of the room, shared with the neighbour behind it (roomgrid.py:285-331): colour and lockedness are
        drawn from the construction-time generator when not given, the position from `np_random` (or the wall's middle)."""
        room = self.get_room(col, row)
        if dir is None:
            while room.neighbors[dir] is None or room.doors[dir] is not None:     # (KeyError on the first look-up: module docstring)
                dir = self._rand_elem(Direction)
        else:
            assert room.neighbors[dir] is not None, "no neighbor in this direction"
            assert room.doors[dir] is None, "door already exists"
        color = color if color is not None else self._rand_color()
        locked = locked if locked is not None else self._rand_bool()
        door = Door(color, is_locked=locked)
        pos = room.set_door_pos(dir, random=self.np_random if rand_pos else None)
        self.put_obj(door, *pos)
        room.doors[dir] = door
        room.neighbors[dir].doors[_opposite(dir)] = door
        return door, pos

    def remove_wall(self, col: int, row: int, dir):
        """Open wall `dir` of the room between its corners; both rooms then count as connected (roomgrid.py:333-374)."""
        room = self.get_room(col, row)
        assert room.doors[dir] is None, "door exists on this wall"
        assert room.neighbors[dir], "invalid wall"
        fixed, lo, hi, vertical = room._wall_span(dir)
        for along in range(lo + 1, hi):
            self.grid.set(*((fixed, along) if vertical else (along, fixed)), None)
        room.doors[dir] = True
        room.neighbors[dir].doors[_opposite(dir)] = True

    def place_agent(self, agent, col: int | None = None, row: int | None = None, rand_dir: bool = True):
        """The agent somewhere in the room (a random one when not given), re-drawn until the cell in front of it is free or a wall
        (roomgrid.py:376-404)."""
        col = col if col is not None else self._rand_int(0, self.num_cols)
        row = row if row is not None else self._rand_int(0, self.num_rows)
        room = self.get_room(col, row)
        while True:
            super().place_agent(agent, room.top, room.size, rand_dir, max_tries=1000)
            ahead = self.grid.get(*agent.front_pos)
            if ahead is None or ahead.type == Type.wall:
                return agent.state.pos

    def connect_all(self, door_colors=list(Color), max_itrs: int = 5000) -> list:
        """Unlocked doors at random walls until every room is reachable from room (0, 0) (roomgrid.py:406-452): three draws per
        iteration (column, row, direction), a fourth for the colour when a door is added."""
        added = []

        def connected(room):
            return [room.neighbors[d] for d in Direction if room.doors[d] is not None]

        for _ in range(max_itrs):
            if len(bfs(self.get_room(0, 0), connected)) == self.num_rows * self.num_cols:
                return added
            col, row = self._rand_int(0, self.num_cols), self._rand_int(0, self.num_rows)
            dir = self._rand_elem(Direction)
            room = self.get_room(col, row)
            other = room.neighbors[dir]
            if not other or room.doors[dir] or room.locked or other.locked:
                continue
            door, _ = self.add_door(col, row, dir=dir, color=self._rand_elem(door_colors), locked=False)
            added.append(door)
        raise RecursionError("connect_all() failed")

    def add_distractors(self, col: int | None = None, row: int | None = None, num_distractors: int = 10,
                        all_unique: bool = True) -> list:
        """Random keys / balls / boxes that are not the (type, colour) of anything already in a room (roomgrid.py:454-495).  As in
        the reference the bookkeeping of seen keys fails after the first one is placed (module docstring)."""
        seen = {(obj.type, obj.color) for rooms in self.room_grid for room in rooms for obj in room.objs}
        placed = []
        while len(placed) < num_distractors:
            color = self._rand_color()
            kind = self._rand_elem([Type.key, Type.ball, Type.box])
            if all_unique and (kind, color) in seen:
                continue
            col = col if col is not None else self._rand_int(0, self.num_cols)
            row = row if row is not None else self._rand_int(0, self.num_rows)
            obj, _ = self.add_object(col, row, kind=kind, color=color)
            seen.append((kind, color))                   # (a set: AttributeError, as roomgrid.py:493)
            placed.append(obj)
        return placed
