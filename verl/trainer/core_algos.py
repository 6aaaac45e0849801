"""RL math entry points with the reference's names (verl/trainer/core_algos.py), computed by the HIP kernels.

  compute_grpo_outcome_advantage  (:137-175)  -> st_grpo_advantage
  compute_policy_loss / compute_kl (:291-353, :394-436) live inside st_grpo_loss, fused with the masked means and the
  gradient w.r.t. the log-probs (see spatialthinker_amd/model.py forward_backward); exposed here for inspection.
  FixedKLController / AdaptiveKLController (:36-89) are plain host objects.
GAE / RLOO / ReMax / REINFORCE++ / value loss are outside the GRPO path and raise."""
from __future__ import annotations

from typing import Tuple

import numpy as np
import torch

from spatialthinker_amd import ops


class FixedKLController:
    def __init__(self, init_kl_coef: float):
        self.kl_coef = init_kl_coef

    def update(self, current_kl: float, n_steps: int) -> None:
        pass


class AdaptiveKLController:
    """https://arxiv.org/pdf/1909.08593.pdf — kl_coef *= 1 + clip(kl/target - 1, -0.2, 0.2) * n_steps / horizon."""

    def __init__(self, init_kl_coef: float, target_kl: float, horizon: float):
        self.kl_coef, self.target, self.horizon = init_kl_coef, target_kl, horizon

    def update(self, current_kl: float, n_steps: int) -> None:
        err = float(np.clip(current_kl / self.target - 1, -0.2, 0.2))
        self.kl_coef *= 1 + err * n_steps / self.horizon


def get_kl_controller(algorithm_config):
    if algorithm_config.kl_type == "fixed":
        return FixedKLController(init_kl_coef=algorithm_config.kl_coef)
    if algorithm_config.kl_type == "adaptive":
        assert algorithm_config.kl_horizon > 0, f"horizon must be larger than 0. Got {algorithm_config.kl_horizon}."
        return AdaptiveKLController(algorithm_config.kl_coef, algorithm_config.kl_target, algorithm_config.kl_horizon)
    raise ValueError(f"Unknown kl type: {algorithm_config.kl_type}.")


@torch.no_grad()
def compute_grpo_outcome_advantage(token_level_rewards: torch.Tensor, response_mask: torch.Tensor, index, eps: float = 1e-6
                                   ) -> Tuple[torch.Tensor, torch.Tensor]:
    """index: per-row group id (uid strings or ints).  Returns (advantages, returns), both (bs, R) on the input device."""
    _, dense = np.unique(np.asarray(index), return_inverse=True)
    dev = torch.device("cuda", torch.cuda.current_device())
    adv, status = ops.grpo_advantage(token_level_rewards.to(dev, torch.float32).contiguous(), response_mask.to(dev, torch.int64).contiguous(),
                                     torch.from_numpy(dense.astype(np.int32)).to(dev), int(dense.max()) + 1, eps)
    if int(status.item()) != 0:
        raise AssertionError("GRPO needs rollout.n > 1.")
    adv = adv.to(token_level_rewards.device)
    return adv, adv


def compute_policy_loss_and_kl(old_log_probs, log_probs, advantages, response_mask, ref_log_probs=None, *, clip_ratio_low=0.2,
                               clip_ratio_high=0.3, clip_ratio_dual=3.0, kl_penalty="low_var_kl", kl_coef=0.0):
    """(pg_loss(+kl_coef*kl), clipfrac_higher, clipfrac_lower, ppo_kl, kl_loss, dloss/dlog_probs) through st_grpo_loss."""
    dev = torch.device("cuda", torch.cuda.current_device())
    f = lambda t: None if t is None else t.reshape(-1).to(dev, torch.float32).contiguous()
    g, met = ops.grpo_loss(f(log_probs), f(old_log_probs), f(ref_log_probs), f(advantages),
                           response_mask.reshape(-1).to(dev, torch.int64).contiguous(), clip_low=clip_ratio_low, clip_high=clip_ratio_high,
                           clip_dual=clip_ratio_dual, kl_kind=kl_penalty, kl_coef=kl_coef, grad_accum=1.0)
    m = met.cpu()
    return m[0], m[1], m[2], m[3], m[5], g.view_as(log_probs)


def _outside_scope(*_a, **_k):
    raise NotImplementedError("only the GRPO estimator is on the SpatialThinker path (algorithm.adv_estimator=grpo)")


compute_gae_advantage_return = compute_rloo_outcome_advantage = compute_remax_outcome_advantage = _outside_scope
compute_reinforce_plus_plus_outcome_advantage = compute_value_loss = _outside_scope
