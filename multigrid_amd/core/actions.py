"""`multigrid.core.actions` of the reference, by name."""
from ..constants import Action  # noqa: F401,F403
