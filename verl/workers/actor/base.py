"""`verl.workers.actor.base.BasePPOActor` — the actor interface a worker drives (reference: verl/workers/actor/base.py:28-66)."""
from abc import ABC, abstractmethod
from typing import Any, Dict

import torch

from ...protocol import DataProto

__all__ = ["BasePPOActor"]


class BasePPOActor(ABC):
    def __init__(self, config):
        self.config = config

    @abstractmethod
    def compute_log_prob(self, data: DataProto) -> torch.Tensor:
        """(bs, response_length) fp32 log-probabilities of the response tokens; `data` carries input_ids / attention_mask / position_ids /
        responses and meta_info["temperature"]."""

    @abstractmethod
    def update_policy(self, data: DataProto) -> Dict[str, Any]:
        """One PPO update over `data` (mini-batches x micro-batches inside); returns the metric lists."""
