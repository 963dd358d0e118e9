"""`multigrid.core.mission` of the reference, by name."""
from ..mission import Mission, MissionSpace  # noqa: F401,F403
