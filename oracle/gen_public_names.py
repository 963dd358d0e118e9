#!/usr/bin/env python3
"""Public attribute names of the reference's data-model classes -> tests/golden/public_names.json.

TEST INFRASTRUCTURE (build container only: needs /root/reference).  Names only -- no reference source -- so that
tests/test_env_compat.py can check that user code written against `multigrid.base.MultiGridEnv`, `multigrid.core.Agent`,
`AgentState`, `Grid` and the WorldObj classes finds the same names on multigrid_amd ("What's missing" #4)."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, os.environ.get("MGX_REFERENCE", "/root/reference"))
sys.path.insert(0, os.path.join(HERE, "standins"))

import numpy as np  # noqa: E402
from multigrid.base import MultiGridEnv  # noqa: E402
from multigrid.core import Agent, AgentState, Grid, WorldObj  # noqa: E402
from multigrid.core.world_object import Box, Door  # noqa: E402


def public(cls, base=object):
    inherited = set(dir(base))
    return sorted(n for n in dir(cls) if not n.startswith("_") and n not in inherited)


from multigrid.core.roomgrid import Room, RoomGrid  # noqa: E402
from multigrid.core.constants import Color, State, Type  # noqa: E402
import multigrid.core.constants as _constants  # noqa: E402
import multigrid.core.roomgrid as _roomgrid  # noqa: E402


def module_names(mod):
    """Public names a module DEFINES or re-exports from this package (not its third-party imports)."""
    return sorted(n for n, v in vars(mod).items() if not n.startswith("_")
                  and (getattr(v, "__module__", "") or "").startswith("multigrid") or n.isupper())


names = {
    # round 6: the room-grid base class, the indexed enums' API (utils/enum.py:42-89, core/constants.py:34-123)
    "Room": public(Room),
    "RoomGrid": public(RoomGrid, MultiGridEnv),
    "IndexedEnum methods": sorted(n for n in vars(_constants.IndexedEnum) if not n.startswith("_")),
    "Color methods": sorted(n for n, v in vars(Color).items() if not n.startswith("_") and not isinstance(v, Color)),
    "Type": public(Type, str),
    "Color": public(Color, str),
    "State": public(State, str),
    "module core.constants": module_names(_constants),
    "module core.roomgrid": [n for n in module_names(_roomgrid) if n in ("Room", "RoomGrid", "bfs", "reject_next_to")],
    "MultiGridEnv": public(MultiGridEnv),
    "Agent": public(Agent),
    "AgentState": public(AgentState, np.ndarray),
    "Grid": public(Grid),
    "WorldObj": public(WorldObj, np.ndarray),
    "Door": public(Door, WorldObj),
    "Box": public(Box, WorldObj),
}
out = os.path.join(os.path.dirname(HERE), "tests", "golden", "public_names.json")
with open(out, "w") as fh:
    json.dump(names, fh, indent=1, sort_keys=True)
print({k: len(v) for k, v in names.items()})
