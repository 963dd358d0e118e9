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


names = {
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
