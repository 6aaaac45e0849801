"""`verl.workers.critic.config` — reference import path (verl/workers/critic/config.py:23-44)."""
from ...trainer.config import CriticConfig, ModelConfig

__all__ = ["CriticConfig", "ModelConfig"]
