"""`multigrid.core.agent` of the reference, by name."""
from ..env import Agent, AgentStateRow as AgentState  # noqa: F401,F403
