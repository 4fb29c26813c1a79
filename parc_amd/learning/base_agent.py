"""Module-path mirror of the reference's learning/base_agent.py: AgentMode and the base class name.  The tracker's
agent stack is implemented as one class in dm_ppo_agent.py."""
from .dm_ppo_agent import AgentMode, DMPPOAgent as BaseAgent  # noqa: F401
