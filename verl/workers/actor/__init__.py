"""`verl.workers.actor` — the reference's package surface (verl/workers/actor/__init__.py:16-29)."""
from .base import BasePPOActor
from .config import ActorConfig, FSDPConfig, ModelConfig, OptimConfig, RefConfig
from .dp_actor import DataParallelPPOActor

__all__ = ["ActorConfig", "BasePPOActor", "DataParallelPPOActor", "FSDPConfig", "ModelConfig", "OptimConfig", "RefConfig"]
