"""`multigrid.core.constants` of the reference, by name."""
from ..constants import *  # noqa: F401,F403
