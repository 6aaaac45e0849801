"""`verl.workers.reward` — the reference's package surface (verl/workers/reward/__init__.py:16-20)."""
from .config import RewardConfig
from .custom import CustomRewardManager

__all__ = ["CustomRewardManager", "RewardConfig"]
