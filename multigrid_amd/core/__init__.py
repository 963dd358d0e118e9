"""`multigrid.core` of the reference, by name (multigrid/core/__init__.py): the classes a user-defined env is written against.

    from multigrid_amd.base import MultiGridEnv
    from multigrid_amd.core import Grid, Goal, Door, Key, Color, Direction

(`import multigrid_amd as multigrid` + the reference's import lines is the drop-in; the submodules `core.grid`, `core.world_object`,
`core.constants`, `core.actions`, `core.agent`, `core.mission` exist under the reference's names too.)
"""
from ..constants import (COLOR_NAMES, COLOR_TO_IDX, COLORS, DIR_TO_VEC, IDX_TO_COLOR, IDX_TO_OBJECT, OBJECT_TO_IDX,  # noqa: F401
                         STATE_TO_IDX, TILE_PIXELS, Action, Color, Direction, IndexedEnum, State, Type)   # (`from .constants import *`)
from ..env import Agent, AgentStateRow as AgentState  # noqa: F401
from ..mission import Mission, MissionSpace  # noqa: F401
from ..world import Ball, Box, Door, Floor, Goal, Grid, Key, Lava, Wall, WorldObj  # noqa: F401

__all__ = ["Action", "Agent", "AgentState", "Color", "Direction", "State", "Type", "DIR_TO_VEC", "COLORS", "COLOR_NAMES", "COLOR_TO_IDX",
           "IDX_TO_COLOR", "IDX_TO_OBJECT", "OBJECT_TO_IDX", "STATE_TO_IDX", "TILE_PIXELS", "IndexedEnum", "Grid", "Mission", "MissionSpace",
           "Ball", "Box", "Door", "Floor", "Goal", "Key", "Lava", "Wall", "WorldObj"]
