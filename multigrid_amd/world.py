"""Host-side world objects and grid: what a user-defined `_gen_grid` is written against.

The reference's extension point is `MultiGridEnv._gen_grid(width, height)` (multigrid/base.py:229-247, called from `reset`,
base.py:280): a subclass builds `self.grid = Grid(width, height)`, draws walls (`wall_rect` / `horz_wall` / `vert_wall`,
multigrid/core/grid.py:133-195), puts `WorldObj`s (`Goal()`, `Door(color, is_locked=True)`, ..., multigrid/core/world_object.py:
279-616) with `put_obj` / `place_obj` and places the agents (base.py:604-697).  Here those classes are light VALUE objects --
a `(type, color, state)` triple with the reference's constructor signatures, properties and predicates -- and `Grid` is a host
array `(W, H, 3)` indexed `[x, y]` like the reference's `Grid.state` (grid.py:54).  Nothing here runs per step: after `_gen_grid`
returns, the grid is checked (outer wall ring, packable values), packed and uploaded, and the HIP kernels own it
(multigrid_amd/env.py: MultiGridEnv.reset).

`Box(color, contains=obj)` (world_object.py:574-605) is carried to the device in the spare bits of the box's cell (include/mgx.h
"BOX CONTENTS": a content kind + colour): a key, ball, goal, floor, lava, wall or a closed unlocked door; toggling the box puts
the content on the grid, as `Box.toggle` does.  Limits, stated where they bite: a box inside a box and a door that is open or
locked inside a box are refused (`content_code`); user-defined object TYPES (the reference's dynamic enum extension,
multigrid/utils/enum.py:51-64) do not exist.
"""
from __future__ import annotations

from typing import Callable

import numpy as np

from .constants import EMPTY_CELL, Color, State, Type


def _index(enum_cls, value, what: str) -> int:
    """An enum member, its integer index or its name (the reference's enums are string-valued: `Color('red')`, `Color.red`)."""
    if isinstance(value, enum_cls):
        return int(value)
    if isinstance(value, (int, np.integer)) and not isinstance(value, bool):
        return int(enum_cls(int(value)))
    name = getattr(value, "value", value)
    if isinstance(name, str):
        try:
            return int(enum_cls[name])
        except KeyError:
            pass
    raise ValueError(f"{value!r} is not a valid {what}")


class WorldObj:
    """A grid cell's content as `(type, color, state)` (multigrid/core/world_object.py:66-137).  Iterates / converts to the three
    integers, so `grid.state[x, y] = obj` works as in the reference (where a WorldObj IS the int array)."""

    TYPE, COLOR, STATE = 0, 1, 2
    dim = 3
    type_name: str | None = None

    def __init__(self, type=None, color=Color.red):
        tname = type if type is not None else (self.type_name or self.__class__.__name__.lower())
        self._v = [_index(Type, tname, "object type"), _index(Color, color, "color"), 0]
        self.contains = None
        self.init_pos: tuple[int, int] | None = None
        self.cur_pos: tuple[int, int] | None = None
        cls = _TYPE_TO_CLASS.get(self._v[0])
        if cls is not None and self.__class__ is WorldObj:
            self.__class__ = cls                                    # WorldObj(type='goal') is a Goal (world_object.py:106-107)

    # -- the triple
    def __iter__(self):
        return iter(self._v)

    def __len__(self):
        return 3

    def __getitem__(self, k):
        return self._v[k]

    def __setitem__(self, k, v):
        self._v[k] = int(v)

    def __array__(self, dtype=None, copy=None):
        return np.asarray(self._v, dtype=dtype or np.int64)

    def __bool__(self):
        return self._v[0] != Type.empty                              # world_object.py:117-118

    def __eq__(self, other):
        return self is other                                         # world_object.py:126-127: identity

    __hash__ = object.__hash__

    def __repr__(self):
        return f"{self.__class__.__name__}(color={self.color.name})"

    @property
    def type(self) -> Type:
        return Type(self._v[0])

    @property
    def color(self) -> Color:
        return Color(self._v[1])

    @color.setter
    def color(self, value):
        self._v[1] = _index(Color, value, "color")

    @property
    def state(self) -> State:
        return State(self._v[2])

    @state.setter
    def state(self, value):
        self._v[2] = _index(State, value, "state")

    def encode(self) -> tuple[int, int, int]:
        return tuple(self._v)

    @staticmethod
    def empty() -> "WorldObj":
        return WorldObj(type=Type.empty)

    @staticmethod
    def from_array(arr) -> "WorldObj | None":
        """world_object.py:139-160: None for an empty cell.  A box cell's state value may carry its content (include/mgx.h "BOX
        CONTENTS": state | kind << 2 | colour << 5): the Box comes back holding it."""
        t = int(arr[0])
        if t == Type.empty:
            return None
        cls = _TYPE_TO_CLASS.get(t)
        if cls is None:
            raise ValueError(f"Unknown object type: {t}")
        obj = cls.__new__(cls)
        WorldObj.__init__(obj, type=Type(t))
        obj._v = [int(arr[0]), int(arr[1]), int(arr[2]) & 3]
        if t == Type.box and int(arr[2]) >> 2:
            obj.contains = content_from_code(int(arr[2]) >> 2)
        return obj

    @staticmethod
    def decode(type_idx: int, color_idx: int, state_idx: int) -> "WorldObj | None":
        return WorldObj.from_array((type_idx, color_idx, state_idx))

    # -- the rule predicates (world_object.py:197-233 + overrides); the device applies the same table (csrc/mgx_rules.h: eval_agent)
    def can_overlap(self) -> bool:
        return self.type == Type.empty

    def can_pickup(self) -> bool:
        return False

    def can_contain(self) -> bool:
        return False

    def toggle(self, env, agent, pos) -> bool:
        """world_object.py:215-233: what the toggle action does to this object (the device applies the same table, csrc/mgx_rules.h:
        eval_agent); here on whatever grid `env.grid` is -- the host grid inside `_gen_grid`, the device-backed view otherwise."""
        return False


class Goal(WorldObj):
    def __init__(self, color=Color.green):                           # world_object.py:284-285
        super().__init__(color=color)

    def can_overlap(self) -> bool:
        return True


class Floor(WorldObj):
    def __init__(self, color=Color.blue):                            # world_object.py:305-306
        super().__init__(color=color)

    def can_overlap(self) -> bool:
        return True


class Lava(WorldObj):
    def __init__(self):                                              # world_object.py:334-337
        super().__init__(color=Color.red)

    def can_overlap(self) -> bool:
        return True


class Wall(WorldObj):
    def __init__(self, color=Color.grey):                            # world_object.py:370-377
        super().__init__(color=color)


class Door(WorldObj):
    """world_object.py:386-474: state open / closed / locked; `is_open` / `is_locked` as the reference's setters combine them."""

    def __init__(self, color=Color.blue, is_open: bool = False, is_locked: bool = False):
        super().__init__(color=color)
        self.is_open = is_open
        self.is_locked = is_locked

    def __repr__(self):
        return f"{self.__class__.__name__}(color={self.color.name},state={self.state.name})"

    @property
    def is_open(self) -> bool:
        return self._v[2] == State.open

    @is_open.setter
    def is_open(self, value: bool):
        if value:
            self._v[2] = int(State.open)
        elif not self.is_locked:
            self._v[2] = int(State.closed)

    @property
    def is_locked(self) -> bool:
        return self._v[2] == State.locked

    @is_locked.setter
    def is_locked(self, value: bool):
        if value:
            self._v[2] = int(State.locked)
        elif not self.is_open:
            self._v[2] = int(State.closed)

    def can_overlap(self) -> bool:
        return self.is_open

    def toggle(self, env, agent, pos) -> bool:
        """world_object.py:458-474"""
        if self.is_locked:
            carried = agent.state.carrying
            if isinstance(carried, Key) and carried.color == self.color:
                self.is_locked = False
                self.is_open = True
                env.grid.update(*pos)
                return True
            return False
        self.is_open = not self.is_open
        env.grid.update(*pos)
        return True


class Key(WorldObj):
    def __init__(self, color=Color.blue):                            # world_object.py:509-510
        super().__init__(color=color)

    def can_pickup(self) -> bool:
        return True


class Ball(WorldObj):
    def __init__(self, color=Color.blue):                            # world_object.py:547-548
        super().__init__(color=color)

    def can_pickup(self) -> bool:
        return True


class Box(WorldObj):
    def __init__(self, color=Color.yellow, contains=None):           # world_object.py:574-585
        super().__init__(color=color)
        if contains is not None:
            content_code(contains)                                   # (refuses what the device's cell cannot hold, by name)
        self.contains = contains

    def can_pickup(self) -> bool:
        return True

    def can_contain(self) -> bool:
        return True

    def toggle(self, env, agent, pos) -> bool:
        """world_object.py:599-605: the box is replaced by what it holds."""
        env.grid.set(*pos, self.contains)
        return True


#: content kinds of include/mgx.h "BOX CONTENTS" (0 = nothing)
_CONTENT_KINDS = (None, Type.key, Type.ball, Type.goal, Type.floor, Type.lava, Type.wall, Type.door)


def content_code(obj) -> int:
    """What a box holds as the device carries it: kind | colour << 3 (include/mgx.h "BOX CONTENTS"); 0 for None."""
    if obj is None:
        return 0
    if not isinstance(obj, WorldObj):
        raise TypeError(f"Box(contains=...) takes a WorldObj or None, got {type(obj).__name__}")
    if obj.type == Type.box:
        raise NotImplementedError("multigrid_amd: a box inside a box is not supported (the device's cell holds ONE level of content: "
                                  "a kind and a colour, include/mgx.h)")
    if obj.type == Type.door and obj.state != State.closed:
        raise NotImplementedError("multigrid_amd: a door inside a box must be closed and unlocked (what Door(color) constructs): the "
                                  "device's cell has no room for the content's state (include/mgx.h)")
    if obj.type not in _CONTENT_KINDS[1:]:
        raise NotImplementedError(f"multigrid_amd: a {obj.type.name} cannot be a box's content (include/mgx.h)")
    return _CONTENT_KINDS.index(obj.type) | (int(obj.color) << 3)


def content_from_code(code: int) -> "WorldObj | None":
    kind, color = code & 7, (code >> 3) & 7
    if kind == 0:
        return None
    t = _CONTENT_KINDS[kind]
    return WorldObj.from_array((int(t), color, int(State.closed) if t == Type.door else 0))


_TYPE_TO_CLASS = {int(Type.goal): Goal, int(Type.floor): Floor, int(Type.lava): Lava, int(Type.wall): Wall, int(Type.door): Door,
                  int(Type.key): Key, int(Type.ball): Ball, int(Type.box): Box}


class Grid:
    """multigrid/core/grid.py:27-195: `state` (W, H, 3) int64 indexed [x, y]; objects are materialised from `state` on `get`
    (grid.py:102-117) and written into it on `set` / `update`."""

    def __init__(self, width: int, height: int):
        assert width >= 3
        assert height >= 3
        self.world_objects: dict[tuple[int, int], WorldObj | None] = {}
        self.state = np.zeros((width, height, WorldObj.dim), dtype=np.int64)
        self.state[...] = EMPTY_CELL

    @property
    def width(self) -> int:
        return self.state.shape[0]

    @property
    def height(self) -> int:
        return self.state.shape[1]

    @property
    def grid(self) -> list:
        return [self.get(i, j) for i in range(self.width) for j in range(self.height)]

    def set(self, x: int, y: int, obj: WorldObj | None):
        self.world_objects[x, y] = obj
        if isinstance(obj, WorldObj):
            self.state[x, y] = obj.encode()
        elif obj is None:
            self.state[x, y] = EMPTY_CELL
        else:
            raise TypeError(f"cannot set grid value to {type(obj)}")

    def get(self, x: int, y: int) -> WorldObj | None:
        if (x, y) not in self.world_objects:
            self.world_objects[x, y] = WorldObj.from_array(self.state[x, y])
        return self.world_objects[x, y]

    def update(self, x: int, y: int):
        obj = self.world_objects.get((x, y))
        if obj is not None:
            self.state[x, y] = obj.encode()

    def horz_wall(self, x: int, y: int, length: int | None = None, obj_type: Callable[[], WorldObj] = Wall):
        length = self.width - x if length is None else length
        self.state[x:x + length, y] = obj_type().encode()           # (writes `state` only, like grid.py:152: SURVEY App. C Q8)

    def vert_wall(self, x: int, y: int, length: int | None = None, obj_type: Callable[[], WorldObj] = Wall):
        length = self.height - y if length is None else length
        self.state[x, y:y + length] = obj_type().encode()

    def wall_rect(self, x: int, y: int, w: int, h: int):
        self.horz_wall(x, y, w)
        self.horz_wall(x, y + h - 1, w)
        self.vert_wall(x, y, h)
        self.vert_wall(x + w - 1, y, h)

    def state_with_contents(self) -> np.ndarray:
        """`state` with every box's content in the upper bits of its state value (include/mgx.h "BOX CONTENTS"): the form grids
        are uploaded in.  Contents are attributes of the objects, so only boxes that were `set` as objects have one."""
        out = self.state.copy()
        for (x, y), obj in self.world_objects.items():
            if isinstance(obj, Box) and obj.contains is not None and int(out[x, y, 0]) == Type.box:
                out[x, y, 2] = (int(out[x, y, 2]) & 3) | (content_code(obj.contains) << 2)
        return out

    def encode(self, vis_mask=None) -> np.ndarray:
        """grid.py:310-330 without agents: the state itself (masked cells -> (0, 0, 0))."""
        out = self.state.copy()
        if vis_mask is not None:
            out[~np.asarray(vis_mask, dtype=bool)] = 0
        return out

    @staticmethod
    def decode(array) -> "tuple[Grid, np.ndarray]":
        """grid.py:327-349: an encoding (W, H, 3) back into a `Grid` + the mask of the cells it shows (type != unseen)."""
        array = np.asarray(array)
        width, height, dim = array.shape
        assert dim == WorldObj.dim
        vis_mask = array[..., WorldObj.TYPE] != int(Type.unseen)
        grid = Grid(width, height)
        grid.state[vis_mask] = array[vis_mask]
        return grid, vis_mask
