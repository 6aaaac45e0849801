"""`verl.workers.actor.dp_actor.DataParallelPPOActor` — the reference's actor class name and methods (verl/workers/actor/dp_actor.py:44-63,
171-210 compute_log_prob, 212-293 update_policy) over this build's engine.

The reference wraps an FSDP `nn.Module` and a torch optimizer; here the weights, gradients and AdamW state live in one `ParamStore` driven
by `spatialthinker_amd.actor.PolicyEngine` (HIP kernels, packed passes), so `actor_module` is a PolicyEngine and `actor_optimizer` is
unused (the engine owns its optimizer; pass None).  `FSDPWorker` calls the engine directly (it also hands over the rollout's prompt K/V);
this class is the same computation behind the reference's signatures for code written against them."""
from typing import Any, Dict, Optional

import torch

from ...protocol import DataProto
from .base import BasePPOActor

__all__ = ["DataParallelPPOActor"]


def _as_dict(data: DataProto) -> Dict[str, Any]:
    d = {k: v for k, v in data.batch.items()}
    d.update(data.non_tensor_batch)
    return d


class DataParallelPPOActor(BasePPOActor):
    def __init__(self, config, actor_module, actor_optimizer: Optional[Any] = None):
        super().__init__(config)
        if not (hasattr(actor_module, "compute_log_prob") and hasattr(actor_module, "store")):
            raise TypeError("actor_module must be a spatialthinker_amd.actor.PolicyEngine (this build has no FSDP nn.Module); "
                            "FSDPWorker.actor / FSDPWorker.ref_policy are such engines")
        self.actor_module = actor_module
        self.actor_optimizer = actor_optimizer

    def compute_log_prob(self, data: DataProto) -> torch.Tensor:
        """dp_actor.py:171-210: micro-batches of config.micro_batch_size_per_device_for_experience rows, no gradients."""
        mb = getattr(self.config, "micro_batch_size_per_device_for_experience", None)
        return self.actor_module.compute_log_prob(_as_dict(data), data.meta_info["temperature"], mb).cpu()

    def update_policy(self, data: DataProto) -> Dict[str, Any]:
        """dp_actor.py:212-293: needs old_log_probs, advantages (and ref_log_probs with use_kl_loss) next to the model inputs."""
        return self.actor_module.update_policy(_as_dict(data), data.meta_info["temperature"])
