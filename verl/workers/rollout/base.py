"""`verl.workers.rollout.base.BaseRollout` — the rollout interface (reference: verl/workers/rollout/base.py:22-27)."""
from abc import ABC, abstractmethod

from ...protocol import DataProto

__all__ = ["BaseRollout"]


class BaseRollout(ABC):
    @abstractmethod
    def generate_sequences(self, prompts: DataProto) -> DataProto:
        """prompts (input_ids / attention_mask / position_ids, optional multi_modal_inputs) -> the (bs * n, prompt + response) batch"""
