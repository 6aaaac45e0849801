"""`verl.workers.config` — the reference's import path of WorkerConfig and the role configs (verl/workers/config.py:18-50); the
dataclasses themselves live in verl/trainer/config.py."""
from ..trainer.config import ActorConfig, CriticConfig, FSDPConfig, ModelConfig, OptimConfig, RefConfig, RewardConfig, RolloutConfig, WorkerConfig

__all__ = ["ActorConfig", "CriticConfig", "FSDPConfig", "ModelConfig", "OptimConfig", "RefConfig", "RewardConfig", "RolloutConfig", "WorkerConfig"]
