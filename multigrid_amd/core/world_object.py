"""`multigrid.core.world_object` of the reference, by name."""
from ..world import Ball, Box, Door, Floor, Goal, Key, Lava, Wall, WorldObj  # noqa: F401,F403
