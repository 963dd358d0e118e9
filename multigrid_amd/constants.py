"""Encodings of the MultiGrid world.

Mirrors multigrid/core/constants.py:34-123, multigrid/utils/enum.py:42-89 and multigrid/core/actions.py:5-15 of the reference.
`Type`, `Color`, `State` are the reference's INDEXED ENUMS with its read-only API -- `Type.wall == 'wall'`, `Type('wall')`,
`Type.wall.value == 'wall'`, `.to_index()`, `Type.from_index(2)` (also on index arrays), `int(Type.wall) == 2`, `Color.red.rgb()`,
`Color.cycle(n)`, usable as keys of name-keyed dicts (`COLORS[Color.red]`) -- over the integer indices the kernels' host side computes
with: a member IS its index (an `int`), so `cells[..., 0] == Type.wall` and `agents[:, 0] = Type.agent` keep working on arrays (the
reference's members are `str`s; `isinstance(Type.wall, str)` is the one thing that differs, and `Type.wall == 2` holds in addition).
`Direction` and `Action` are the reference's IntEnums.  The reference's dynamic enum extension (`IndexedEnum.add_item`,
`Color.add_color`: multigrid/utils/enum.py:51-64) is out of scope: the kernels hard-code these tables.
"""
from __future__ import annotations

import enum
from types import DynamicClassAttribute

import numpy as np

#: multigrid/core/constants.py:12-19
COLORS = {
    'red': np.array([255, 0, 0]),
    'green': np.array([0, 255, 0]),
    'blue': np.array([0, 0, 255]),
    'purple': np.array([112, 39, 195]),
    'yellow': np.array([255, 255, 0]),
    'grey': np.array([100, 100, 100]),
}
#: multigrid/core/constants.py:9
TILE_PIXELS = 32


class IndexedEnum(enum.IntEnum):
    """multigrid/utils/enum.py:42-89: an enum whose members have a string value (their name) and an integer index.  Here the
    member is the index (module docstring); everything the reference's members answer is answered the same way."""

    @classmethod
    def _missing_(cls, value):
        # Type('wall') (the reference's by-value lookup); a member of the reference-style string kind compares by name too
        name = getattr(value, "value", value)
        if isinstance(name, str) and name in cls.__members__:
            return cls.__members__[name]
        return None

    @DynamicClassAttribute
    def value(self) -> str:                                # constants.py:38-48: the values are the names
        return self._name_

    def __eq__(self, other):
        if isinstance(other, str):
            return self._name_ == other
        return int.__eq__(self, other)

    def __ne__(self, other):
        r = self.__eq__(other)
        return r if r is NotImplemented else not r

    def __hash__(self):                                    # == hash(name): a member finds its entry in a name-keyed dict
        return hash(self._name_)

    def __bool__(self):                                    # (a non-empty string in the reference: `color or self._rand_color()`,
        return True                                        # roomgrid.py:281-282, keeps Color.red)

    def __str__(self):                                     # (the reference: `str`-mixin enums print as 'Type.wall')
        return f"{self.__class__.__name__}.{self._name_}"

    def __format__(self, spec):                            # f"{Color.red}" -> 'red', as a str-mixin enum formats (py < 3.12)
        return format(self._name_, spec)

    def to_index(self) -> int:                             # enum.py:86-89
        return int(self)

    @classmethod
    def from_index(cls, index):                            # enum.py:66-84: a member, or the array of VALUES for an index array
        names = np.array([m._name_ for m in cls])
        out = names[np.asarray(index)]
        return cls.__members__[str(out)] if out.ndim == 0 else out

    @classmethod
    def add_item(cls, name, value):                        # enum.py:51-64
        raise NotImplementedError("multigrid_amd: the object tables are compiled into the kernels; dynamic enum extension "
                                  "(multigrid/utils/enum.py:51-64) is out of scope")


class Type(IndexedEnum):
    unseen = 0
    empty = 1
    wall = 2
    floor = 3
    door = 4
    key = 5
    ball = 6
    box = 7
    goal = 8
    lava = 9
    agent = 10


class Color(IndexedEnum):
    red = 0
    green = 1
    blue = 2
    purple = 3
    yellow = 4
    grey = 5

    @classmethod
    def add_color(cls, name, rgb):                         # constants.py:63-75
        cls.add_item(name, name)

    @staticmethod
    def cycle(n: int) -> tuple["Color", ...]:
        """multigrid/core/constants.py:77-82"""
        return tuple(Color(i % len(Color)) for i in range(int(n)))

    def rgb(self) -> np.ndarray:
        """multigrid/core/constants.py:84-88"""
        return COLORS[self]


class State(IndexedEnum):
    open = 0
    closed = 1
    locked = 2


class Direction(enum.IntEnum):
    right = 0
    down = 1
    left = 2
    up = 3

    def to_vec(self) -> np.ndarray:
        return DIR_TO_VEC[self]


class Action(enum.IntEnum):
    left = 0
    right = 1
    forward = 2
    pickup = 3
    drop = 4
    toggle = 5
    done = 6


#: multigrid/core/constants.py:21-30
DIR_TO_VEC = [np.array((1, 0)), np.array((0, 1)), np.array((-1, 0)), np.array((0, -1))]

#: MiniGrid compatibility tables (multigrid/core/constants.py:117-123)
OBJECT_TO_IDX = {t: t.to_index() for t in Type}
IDX_TO_OBJECT = {t.to_index(): t for t in Type}
COLOR_TO_IDX = {c: c.to_index() for c in Color}
IDX_TO_COLOR = {c.to_index(): c for c in Color}
STATE_TO_IDX = {s: s.to_index() for s in State}
COLOR_NAMES = sorted(Color, key=lambda c: c.value)

#: cell encodings (type, color, state)
EMPTY_CELL = (int(Type.empty), 0, 0)            # multigrid/core/world_object.py:131-137
WALL_CELL = (int(Type.wall), int(Color.grey), 0)  # multigrid/utils/obs.py:14
UNSEEN_CELL = (0, 0, 0)                          # multigrid/utils/obs.py:15
GOAL_CELL = (int(Type.goal), int(Color.green), 0)  # multigrid/core/world_object.py:284-285
LAVA_CELL = (int(Type.lava), int(Color.red), 0)    # multigrid/core/world_object.py:334-337

#: action value meaning "this agent is absent from the actions dict" (multigrid/base.py:403-404)
NO_ACTION = -1

#: packed agent row layout (include/mgx.h)
AG_COLOR, AG_DIR, AG_X, AG_Y, AG_TERMINATED, AG_CARRY = 0, 1, 2, 3, 4, 5
AGENT_STRIDE = 8
