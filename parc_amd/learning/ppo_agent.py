"""Module-path mirror of the reference's learning/ppo_agent.py (see dm_ppo_agent.py)."""
from .dm_ppo_agent import DMPPOAgent as PPOAgent  # noqa: F401
