"""`verl.workers.critic.dp_critic.DataParallelPPOCritic` — the reference's critic class name and methods (verl/workers/critic/dp_critic.py:
40-50, 127-152 compute_values, 154-225 update_critic) over `spatialthinker_amd.actor.CriticEngine` (see dp_actor.py for why the module argument
is an engine and the optimizer argument is unused)."""
from typing import Any, Dict, Optional

import torch

from ...protocol import DataProto
from ..actor.dp_actor import _as_dict
from .base import BasePPOCritic

__all__ = ["DataParallelPPOCritic"]


class DataParallelPPOCritic(BasePPOCritic):
    def __init__(self, config, critic_module, critic_optimizer: Optional[Any] = None):
        super().__init__(config)
        if not (hasattr(critic_module, "compute_values") and hasattr(critic_module, "update_critic")):
            raise TypeError("critic_module must be a spatialthinker_amd.actor.CriticEngine (FSDPWorker.critic in the `critic` role)")
        self.critic_module = critic_module
        self.critic_optimizer = critic_optimizer

    def compute_values(self, data: DataProto) -> torch.Tensor:
        return self.critic_module.compute_values(_as_dict(data)).cpu()

    def update_critic(self, data: DataProto) -> Dict[str, Any]:
        return self.critic_module.update_critic(_as_dict(data))
