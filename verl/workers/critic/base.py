"""`verl.workers.critic.base.BasePPOCritic` — the critic interface (reference: verl/workers/critic/base.py:28-42)."""
from abc import ABC, abstractmethod
from typing import Any, Dict

import torch

from ...protocol import DataProto

__all__ = ["BasePPOCritic"]


class BasePPOCritic(ABC):
    def __init__(self, config):
        self.config = config

    @abstractmethod
    def compute_values(self, data: DataProto) -> torch.Tensor:
        """(bs, response_length) fp32 value predictions, zero outside the response mask."""

    @abstractmethod
    def update_critic(self, data: DataProto) -> Dict[str, Any]:
        """One critic update over `data`; returns the metric lists."""
