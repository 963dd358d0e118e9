"""Concrete environments + the name -> (class, kwargs) table of the reference.

* `EmptyEnv`                   multigrid/envs/empty.py:15-170
* `BlockedUnlockPickupEnv`     multigrid/envs/blockedunlockpickup.py:12-175 (over RoomGrid, core/roomgrid.py:139-236)
* `CONFIGURATIONS`, `make`     multigrid/envs/__init__.py:38-57 (the in-scope subset; the reference registers these
                               ids with gymnasium, which is not a dependency here -- `make(id, **kwargs)` plays the
                               part of `gym.make`)
"""
from __future__ import annotations

from . import layouts
from .constants import Color, Direction, Type
from .env import MultiGridEnv
from .mission import MissionSpace
from .spec import EnvSpec


class EmptyEnv(MultiGridEnv):
    env_kind = "empty"

    def __init__(self, size: int = 8, agent_start_pos: tuple[int, int] | None = (1, 1),
                 agent_start_dir: Direction | None = Direction.right, max_steps: int | None = None,
                 joint_reward: bool = False, success_termination_mode: str = "any", **kwargs):
        self.agent_start_pos = agent_start_pos
        self.agent_start_dir = agent_start_dir
        super().__init__(mission_space="get to the green goal square", grid_size=size,
                         max_steps=max_steps or (4 * size ** 2),                   # empty.py:145
                         joint_reward=joint_reward, success_termination_mode=success_termination_mode, **kwargs)

    def _gen_layout(self, layout_rng, np_random):
        grid, agents = layouts.empty_layout(self.width, self.num_agents, self.agent_start_pos,
                                            self.agent_start_dir, layout_rng)
        return grid, agents, None


class BlockedUnlockPickupEnv(MultiGridEnv):
    env_kind = "blockedunlockpickup"

    def __init__(self, room_size: int = 6, max_steps: int | None = None, joint_reward: bool = True, **kwargs):
        assert room_size >= 4                                                         # blockedunlockpickup.py:120
        self.room_size = room_size
        mission_space = MissionSpace(mission_func=self._gen_mission,
                                     ordered_placeholders=[[c.name for c in Color], ["box", "key"]])
        super().__init__(mission_space=mission_space, width=(room_size - 1) * 2 + 1, height=room_size,
                         max_steps=max_steps or (16 * room_size ** 2),                # blockedunlockpickup.py:132
                         joint_reward=joint_reward, success_termination_mode="any", **kwargs)

    @staticmethod
    def _gen_mission(color: str, obj_type: str):
        return f"pick up the {color} {obj_type}"

    def _gen_layout(self, layout_rng, np_random):
        grid, agents, target = layouts.blockedunlockpickup_layout(self.room_size, self.num_agents, layout_rng,
                                                                  np_random)
        # blockedunlockpickup.py:164 (the agents' obs keep the sampled mission)
        self.mission = f"pick up the {Color(int(target[1])).name} {Type(int(target[0])).name}"
        return grid, agents, target


#: multigrid/envs/__init__.py:38-52, restricted to the env classes in scope
CONFIGURATIONS = {
    "MultiGrid-BlockedUnlockPickup-v0": (BlockedUnlockPickupEnv, {}),
    "MultiGrid-Empty-5x5-v0": (EmptyEnv, {"size": 5}),
    "MultiGrid-Empty-Random-5x5-v0": (EmptyEnv, {"size": 5, "agent_start_pos": None}),
    "MultiGrid-Empty-6x6-v0": (EmptyEnv, {"size": 6}),
    "MultiGrid-Empty-Random-6x6-v0": (EmptyEnv, {"size": 6, "agent_start_pos": None}),
    "MultiGrid-Empty-8x8-v0": (EmptyEnv, {}),
    "MultiGrid-Empty-16x16-v0": (EmptyEnv, {"size": 16}),
}


def make(env_id: str, **kwargs) -> MultiGridEnv:
    """`gym.make(env_id, **kwargs)` for the ids above."""
    if env_id not in CONFIGURATIONS:
        raise KeyError(f"unknown environment id {env_id!r}; available: {sorted(CONFIGURATIONS)}")
    cls, cfg = CONFIGURATIONS[env_id]
    return cls(**{**cfg, **kwargs})


def spec_for(env_id: str, agents: int = 1, **kwargs) -> EnvSpec:
    """The EnvSpec `make(env_id, agents=..., **kwargs)` would run with, without touching a device
    (for `BatchedMultiGridEnv` users)."""
    cls, cfg = CONFIGURATIONS[env_id]
    kw = {**cfg, **kwargs}
    common = dict(
        num_agents=agents, view_size=kw.get("agent_view_size", 7),
        see_through_walls=kw.get("see_through_walls", False),
        allow_agent_overlap=kw.get("allow_agent_overlap", True),
        failure_termination_mode=kw.get("failure_termination_mode", "all"))
    if cls is EmptyEnv:
        size = kw.get("size", 8)
        return EnvSpec(width=size, height=size, max_steps=kw.get("max_steps") or 4 * size ** 2,
                       joint_reward=kw.get("joint_reward", False),
                       success_termination_mode=kw.get("success_termination_mode", "any"),
                       env_kind="empty", **common)
    rs = kw.get("room_size", 6)
    return EnvSpec(width=(rs - 1) * 2 + 1, height=rs, max_steps=kw.get("max_steps") or 16 * rs ** 2,
                   joint_reward=kw.get("joint_reward", True), success_termination_mode="any",
                   env_kind="blockedunlockpickup", **common)
