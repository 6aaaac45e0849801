"""`verl.workers.reward.config` — reference import path (verl/workers/reward/config.py:21-25)."""
from ...trainer.config import RewardConfig

__all__ = ["RewardConfig"]
