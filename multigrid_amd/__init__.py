"""multigrid_amd -- MI355X-native batched MultiGrid step/observation engine.

A drop-in for the hot path of ini/multigrid (`MultiGridEnv.step` / `gen_obs`): thousands of independent
gridworlds live in HBM as uint8 tensors and are stepped by one fused hand-written HIP kernel (gfx950).
See DESIGN.md for the data layout and kernels, INTEGRATION.md for the C ABI, include/mgx.h for signatures.
"""
from .constants import Action, Color, Direction, State, Type  # noqa: F401
from .spec import EnvSpec  # noqa: F401
from .batched import BatchedMultiGridEnv  # noqa: F401

__all__ = ["Action", "Color", "Direction", "State", "Type", "EnvSpec", "BatchedMultiGridEnv"]
from .env import Agent, MultiGridEnv  # noqa: F401,E402
from .envs import (CONFIGURATIONS, BlockedUnlockPickupEnv, EmptyEnv, LockedHallwayEnv, PlaygroundEnv,  # noqa: F401,E402
                   RedBlueDoorsEnv, make, spec_for)
from .rllib import RLlibWrapper, to_rllib_env  # noqa: F401,E402

__all__ += ["Agent", "MultiGridEnv", "CONFIGURATIONS", "BlockedUnlockPickupEnv", "EmptyEnv", "LockedHallwayEnv", "PlaygroundEnv",
            "RedBlueDoorsEnv", "make", "spec_for",
            "RLlibWrapper", "to_rllib_env"]
from .wrappers import FullyObsWrapper, ImgObsWrapper, OneHotObsWrapper, SingleAgentWrapper  # noqa: F401,E402

__all__ += ["FullyObsWrapper", "ImgObsWrapper", "OneHotObsWrapper", "SingleAgentWrapper"]
from .world import Ball, Box, Door, Floor, Goal, Grid, Key, Lava, Wall, WorldObj  # noqa: F401,E402
from .mission import MissionSpace  # noqa: F401,E402

__all__ += ["Ball", "Box", "Door", "Floor", "Goal", "Grid", "Key", "Lava", "Wall", "WorldObj", "MissionSpace"]
