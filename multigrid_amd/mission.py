"""Mission strings (host-side pass-through; multigrid/core/mission.py:9-136 of the reference)."""
from __future__ import annotations

from typing import Callable, Iterable, Sequence

import numpy as np

from .spaces import MultiDiscrete


class Mission(str):
    """A mission string.  The reference's `Mission` is an ndarray subclass carrying `.string`
    (mission.py:9-42); equality and `str()` are by text, which is what this class keeps."""

    def __new__(cls, string: str, index: Iterable[int] | None = None):
        obj = super().__new__(cls, string)
        obj.index = None if index is None else tuple(int(i) for i in index)
        return obj

    @property
    def string(self) -> str:
        return str.__str__(self)

    def __repr__(self) -> str:
        return f'Mission("{self.string}")'


class MissionSpace(MultiDiscrete):
    """multigrid/core/mission.py:45-136"""

    def __init__(self, mission_func: Callable[..., str], ordered_placeholders: Sequence[Sequence[str]] = ()):
        self.mission_func = mission_func
        self.arg_groups = [list(g) for g in ordered_placeholders]
        nvec = tuple(len(group) for group in self.arg_groups)
        super().__init__(nvec if nvec else (1,))

    def get(self, idx: Iterable[int]) -> Mission:
        if self.arg_groups:
            args = (self.arg_groups[axis][int(index)] for axis, index in enumerate(idx))
            return Mission(self.mission_func(*args), index=idx)
        return Mission(self.mission_func())

    def sample(self) -> Mission:
        return self.get(super().sample())

    def contains(self, x) -> bool:
        return any(self.get(idx) == x for idx in np.ndindex(tuple(int(n) for n in self.nvec)))

    @staticmethod
    def from_string(string: str) -> "MissionSpace":
        return MissionSpace(mission_func=lambda: string)

    def __repr__(self) -> str:
        if self.arg_groups:
            return f"MissionSpace({self.mission_func.__name__}, {self.arg_groups})"
        return f"MissionSpace('{self.mission_func()}')"
