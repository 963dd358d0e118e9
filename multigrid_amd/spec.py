"""EnvSpec: the constructor-level configuration of one environment class, frozen.

Collects the keyword arguments of `MultiGridEnv.__init__` (multigrid/base.py:85-103) that drive the hot path,
plus `env_kind`, which selects the env-specific post-step hook (multigrid/envs/blockedunlockpickup.py:166-175).
"""
from __future__ import annotations

import ctypes as C
import dataclasses
from dataclasses import dataclass

ENV_KINDS = {"empty": 0, "blockedunlockpickup": 1, "redbluedoors": 2, "lockedhallway": 3, "rules": 4}
AUX_BYTES = 16
MAX_AGENTS = 32
MAX_VIEW = 15


class MgxSpecC(C.Structure):
    """`struct MgxSpec` of include/mgx.h."""
    _fields_ = [(n, C.c_int32) for n in (
        "width", "height", "num_agents", "view_size", "max_steps", "see_through_walls",
        "allow_agent_overlap", "joint_reward", "success_any", "failure_any", "env_kind", "cell_bytes")]


@dataclass(frozen=True)
class EnvSpec:
    width: int
    height: int
    num_agents: int = 1
    view_size: int = 7                       # multigrid/base.py:93 agent_view_size
    max_steps: int = 100                     # multigrid/base.py:91
    see_through_walls: bool = False          # multigrid/base.py:92
    allow_agent_overlap: bool = True         # multigrid/base.py:95
    joint_reward: bool = False               # multigrid/base.py:96
    success_termination_mode: str = "any"    # multigrid/base.py:97
    failure_termination_mode: str = "all"    # multigrid/base.py:98
    env_kind: str = "empty"
    #: bytes per grid cell on the device (include/mgx.h): 2 = MgxCell (16 bits, every entry point), 1 = MgxCell8 (compact cells for
    #: LARGE grids -- type and state coded jointly in one byte: a third less traffic per step and twice the envs per wavefront on
    #: a 64x64 grid; served by step / gen_obs / auto-reset from a layout pool / full_obs, not by rollouts, one-hot output, device
    #: generation or persistent stepping); 3 = the grid tensor IS the reference's `(type, color, state)` bytes u8[B,H,W,3], packed by
    #: the step kernel itself as it loads them (for callers that hold their state in that form; same set of entry points as 1).
    #: Same results bit for bit either way.
    cell_bytes: int = 2

    def __post_init__(self):
        # multigrid/core/agent.py:78-79
        assert self.view_size % 2 == 1
        assert self.view_size >= 3
        # multigrid/core/grid.py:51-52
        assert self.width >= 3
        assert self.height >= 3
        if self.success_termination_mode not in ("any", "all"):
            raise ValueError(f"success_termination_mode: {self.success_termination_mode!r}")
        if self.failure_termination_mode not in ("any", "all"):
            raise ValueError(f"failure_termination_mode: {self.failure_termination_mode!r}")
        if self.env_kind not in ENV_KINDS:
            raise ValueError(f"env_kind: {self.env_kind!r}")
        if not isinstance(self.max_steps, int):
            raise AssertionError(f"The argument max_steps must be an integer, got: {type(self.max_steps)}")
        if not 1 <= self.num_agents <= MAX_AGENTS:
            raise ValueError(f"num_agents must be in 1..{MAX_AGENTS}")
        if self.view_size > MAX_VIEW:
            raise ValueError(f"view_size must be <= {MAX_VIEW}")
        if self.width > 255 or self.height > 255:
            raise ValueError("grid sides must be <= 255 (positions are stored as uint8)")
        if self.cell_bytes not in (1, 2, 3):
            raise ValueError("cell_bytes must be 2 (MgxCell), 1 (MgxCell8, compact cells) or 3 (the reference's byte triples)")

    # ---- shapes of the device tensors (include/mgx.h) ----
    def grid_shape(self, batch: int):
        """(type, color, state) bytes per cell: the form grids are given in and read back as"""
        return (batch, self.height, self.width, 3)

    def cells_shape(self, batch: int):
        """packed cells: the form the device holds (include/mgx.h MgxCell, MgxCell8 when cell_bytes == 1, the byte triples when 3)"""
        return (batch, self.height, self.width) + ((3,) if self.cell_bytes == 3 else ())

    @property
    def compact(self) -> bool:
        """a cell format that only the plain step / gen_obs kernels serve (compact cells, byte grids)"""
        return self.cell_bytes != 2

    def agents_shape(self, batch: int):
        return (batch, self.num_agents, 8)

    def obs_shape(self, batch: int):
        return (batch, self.num_agents, self.view_size, self.view_size, 3)

    def as_dict(self) -> dict:
        return dataclasses.asdict(self)

    @staticmethod
    def from_dict(d: dict) -> "EnvSpec":
        names = {f.name for f in dataclasses.fields(EnvSpec)}
        return EnvSpec(**{k: v for k, v in d.items() if k in names})

    def to_c(self) -> MgxSpecC:
        return MgxSpecC(
            width=self.width, height=self.height, num_agents=self.num_agents, view_size=self.view_size,
            max_steps=self.max_steps, see_through_walls=int(self.see_through_walls),
            allow_agent_overlap=int(self.allow_agent_overlap), joint_reward=int(self.joint_reward),
            success_any=int(self.success_termination_mode == "any"),
            failure_any=int(self.failure_termination_mode == "any"),
            env_kind=ENV_KINDS[self.env_kind], cell_bytes=self.cell_bytes)

    # ---- algorithmic HBM bytes per agent-step ----
    def bytes_gen_obs(self) -> int:
        v2 = 3 * self.view_size ** 2
        return v2 + min(v2, (3 * self.height * self.width) // self.num_agents) + 16 + 1

    def bytes_step(self) -> int:
        return self.bytes_gen_obs() + 28
