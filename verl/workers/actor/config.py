"""`verl.workers.actor.config` — reference import path (verl/workers/actor/config.py:23-97) of the actor-side config dataclasses."""
from ...trainer.config import ActorConfig, FSDPConfig, ModelConfig, OffloadConfig, OptimConfig, RefConfig

__all__ = ["ActorConfig", "FSDPConfig", "ModelConfig", "OffloadConfig", "OptimConfig", "RefConfig"]
