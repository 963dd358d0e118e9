"""`multigrid.base` of the reference, by name: `from multigrid_amd.base import MultiGridEnv` (multigrid/base.py:36)."""
from .env import MultiGridEnv  # noqa: F401

__all__ = ["MultiGridEnv"]
