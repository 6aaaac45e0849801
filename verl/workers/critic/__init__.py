"""`verl.workers.critic` — the reference's package surface (verl/workers/critic/__init__.py:16-21)."""
from .base import BasePPOCritic
from .config import CriticConfig, ModelConfig
from .dp_critic import DataParallelPPOCritic

__all__ = ["BasePPOCritic", "CriticConfig", "DataParallelPPOCritic", "ModelConfig"]
