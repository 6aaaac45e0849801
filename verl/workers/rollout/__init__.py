"""`verl.workers.rollout` — RolloutConfig and the rollout interface under the reference's import paths (verl/workers/rollout/__init__.py:16-20;
its vLLMRollout has no counterpart class: generation is `spatialthinker_amd.rollout.Generator`, driven by FSDPWorker.generate_sequences)."""
from .base import BaseRollout
from .config import RolloutConfig
from .hip_rollout import assemble_rollout_batch  # noqa: F401

__all__ = ["BaseRollout", "RolloutConfig", "assemble_rollout_batch"]
