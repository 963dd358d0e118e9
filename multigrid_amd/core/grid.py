"""`multigrid.core.grid` of the reference, by name."""
from ..world import Grid  # noqa: F401,F403
