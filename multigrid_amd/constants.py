"""Integer encodings of the MultiGrid world.

Mirrors (values only) multigrid/core/constants.py:34-107 and multigrid/core/actions.py:5-15 of the reference:
`Type`, `Color`, `State` are the *indices* of the reference's string enums (``Type.wall.to_index() == 2``),
`Direction` and `Action` are the reference's IntEnums.  The reference's dynamic enum extension
(multigrid/utils/enum.py:51-64) is out of scope: the kernels hard-code these tables.
"""
from __future__ import annotations

import enum

import numpy as np


class Type(enum.IntEnum):
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


class Color(enum.IntEnum):
    red = 0
    green = 1
    blue = 2
    purple = 3
    yellow = 4
    grey = 5

    @staticmethod
    def cycle(n: int) -> tuple["Color", ...]:
        """multigrid/core/constants.py:77-82"""
        return tuple(Color(i % len(Color)) for i in range(int(n)))


class State(enum.IntEnum):
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
