"""`verl.workers.rollout.config` — reference import path (verl/workers/rollout/config.py:22-46)."""
from ...trainer.config import RolloutConfig

__all__ = ["RolloutConfig"]
