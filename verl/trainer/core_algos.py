"""RL math entry points with the reference's names (verl/trainer/core_algos.py), computed by the HIP kernels.

  compute_grpo_outcome_advantage  (:137-175)  -> st_grpo_advantage
  compute_policy_loss / compute_kl (:291-353, :394-436) live inside st_grpo_loss, fused with the masked means and the
  gradient w.r.t. the log-probs (see spatialthinker_amd/model.py forward_backward); exposed here for inspection.
  FixedKLController / AdaptiveKLController (:36-89) are plain host objects.
The other estimators of the reference — GAE (:92-133), RLOO (:178-215), REINFORCE++ (:218-246), ReMax (:249-278) — and the
clipped value loss (:356-391) are a few elementwise passes over (bs, R) host tensors that the reference also runs on the driver
CPU: plain torch here, pinned by tests/golden/rl_extra.npz.  Batch-wide statistics (masked_whiten) take an optional
`all_reduce` callable so that several SPMD ranks whiten with the statistics of the GLOBAL batch, as the reference's single
driver does."""
from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Tuple

import numpy as np
import torch

from spatialthinker_amd import ops


class KLController(ABC):
    """kl_coef + update(current_kl, n_steps) (core_algos.py:36-43)."""
    kl_coef: float

    @abstractmethod
    def update(self, current_kl: float, n_steps: int) -> None: ...


class FixedKLController(KLController):
    def __init__(self, init_kl_coef: float):
        self.kl_coef = init_kl_coef

    def update(self, current_kl: float, n_steps: int) -> None:
        pass


class AdaptiveKLController(KLController):
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


class _PolicyLoss(torch.autograd.Function):
    """st_grpo_loss with the KL term off = compute_policy_loss: the kernel returns the four masked means and d(pg_loss)/d(log_probs)."""

    @staticmethod
    def forward(ctx, log_probs, old_log_probs, advantages, response_mask, lo, hi, dual):
        dev = log_probs.device if log_probs.is_cuda else torch.device("cuda", torch.cuda.current_device())
        f = lambda t: t.detach().reshape(-1).to(dev, torch.float32).contiguous()
        g, met = ops.grpo_loss(f(log_probs), f(old_log_probs), None, f(advantages), response_mask.reshape(-1).to(dev, torch.int64).contiguous(),
                               clip_low=lo, clip_high=hi, clip_dual=dual, kl_kind="kl", kl_coef=0.0, grad_accum=1.0)
        ctx.save_for_backward(g.view(log_probs.shape).to(log_probs.device))
        m = met.to(log_probs.device)
        outs = [m[i].clone() for i in range(4)]
        ctx.mark_non_differentiable(*outs[1:])               # on the tensors actually returned: the three statistics carry no grad_fn
        return tuple(outs)

    @staticmethod
    def backward(ctx, g_loss, *_):
        (g,) = ctx.saved_tensors
        return g_loss * g, None, None, None, None, None, None


def compute_policy_loss(old_log_probs: torch.Tensor, log_probs: torch.Tensor, advantages: torch.Tensor, response_mask: torch.Tensor,
                        clip_ratio_low: float, clip_ratio_high: float, clip_ratio_dual: float
                        ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """core_algos.py:291-353 under the reference's name and argument order: (pg_loss, pg_clipfrac_higher, pg_clipfrac_lower, ppo_kl),
    all token-means over response_mask (masked_mean, eps 1e-8); pg_loss is differentiable w.r.t. log_probs (dual-clip PPO:
    max(-A r, -A clip(r)) capped at -A * clip_ratio_dual where A < 0).  One launch of st_grpo_loss."""
    return _PolicyLoss.apply(log_probs, old_log_probs, advantages, response_mask, float(clip_ratio_low), float(clip_ratio_high), float(clip_ratio_dual))


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


# ------------------------------------------------------------------ other estimators (host math)
def masked_mean(values: torch.Tensor, mask: torch.Tensor, dim=None, eps: float = 1e-8) -> torch.Tensor:
    """VF.masked_mean (verl/utils/torch_functional.py:69-71)."""
    return (values * mask).sum(dim=dim) / (mask.sum(dim=dim) + eps)


def masked_whiten(values: torch.Tensor, mask: torch.Tensor, eps: float = 1e-8, all_reduce=None) -> torch.Tensor:
    """VF.masked_whiten (torch_functional.py:74-97): (v - mean) * rsqrt(var_unbiased + eps) over the masked entries.
    all_reduce(t) -> t summed over the ranks (in place or not): the statistics then cover every rank's rows."""
    red = (lambda t: t) if all_reduce is None else all_reduce
    n = red(mask.sum().to(torch.float32).reshape(1).clone())[0]
    mean = red((values * mask).sum().reshape(1).clone())[0] / (n + 1e-8)
    var = red((((values - mean) ** 2) * mask).sum().reshape(1).clone())[0] / (n + 1e-8)
    if n > 1:
        var = var * (n / (n - 1))
    else:
        print("The sum of the mask is less than one, which can cause a division by zero.")
    return (values - mean) * torch.rsqrt(var + eps)


@torch.no_grad()
def compute_gae_advantage_return(token_level_rewards, values, response_mask, gamma, lam, all_reduce=None):
    """A_t = delta_t + gamma*lam*A_{t+1}, delta_t = r_t + gamma*V_{t+1} - V_t (V past the end = 0); returns = A + V; A whitened."""
    R = token_level_rewards.shape[-1]
    adv = torch.zeros_like(token_level_rewards)
    carry = torch.zeros_like(token_level_rewards[:, 0])
    for t in range(R - 1, -1, -1):
        nxt = values[:, t + 1] if t + 1 < R else 0.0
        carry = token_level_rewards[:, t] + gamma * nxt - values[:, t] + gamma * lam * carry
        adv[:, t] = carry
    returns = adv + values
    return masked_whiten(adv, response_mask, all_reduce=all_reduce), returns


@torch.no_grad()
def compute_rloo_outcome_advantage(token_level_rewards, response_mask, index):
    """Leave-one-out baseline inside each uid group: A_i = s_i - (sum_g - s_i) / (n_g - 1)."""
    scores = token_level_rewards.sum(-1)
    _, dense = np.unique(np.asarray(index), return_inverse=True)
    g = torch.from_numpy(dense.astype(np.int64))
    n_groups = int(dense.max()) + 1
    cnt = torch.zeros(n_groups, dtype=scores.dtype).index_add_(0, g, torch.ones_like(scores))
    if bool((cnt < 2).any()):
        raise AssertionError("RLOO needs rollout.n > 1.")
    tot = torch.zeros(n_groups, dtype=scores.dtype).index_add_(0, g, scores)
    out = scores - (tot[g] - scores) / (cnt[g] - 1)
    ret = out.unsqueeze(-1) * response_mask
    return ret, ret


@torch.no_grad()
def compute_reinforce_plus_plus_outcome_advantage(token_level_rewards, response_mask, gamma, all_reduce=None):
    """Discounted return-to-go, restarted behind the EOS, whitened over the batch."""
    returns = torch.zeros_like(token_level_rewards)
    run = torch.zeros_like(token_level_rewards[:, 0])
    for t in range(token_level_rewards.shape[1] - 1, -1, -1):
        run = token_level_rewards[:, t] + gamma * run
        returns[:, t] = run
        run = run * response_mask[:, t]
    return masked_whiten(returns, response_mask, all_reduce=all_reduce), returns


@torch.no_grad()
def compute_remax_outcome_advantage(token_level_rewards, reward_baselines, response_mask):
    """Outcome score minus the score of the greedy rollout of the same prompt."""
    ret = (token_level_rewards.sum(-1) - reward_baselines).unsqueeze(-1) * response_mask
    return ret, ret


def compute_value_loss(vpreds, returns, values, action_mask, cliprange_value: float):
    """0.5 * masked_mean(max((v - R)^2, (clip(v, V_old +- c) - R)^2)) and the fraction of clipped entries."""
    clipped = torch.clamp(vpreds, values - cliprange_value, values + cliprange_value)
    l1, l2 = (vpreds - returns) ** 2, (clipped - returns) ** 2
    return 0.5 * masked_mean(torch.max(l1, l2), action_mask), masked_mean((l1 < l2).float(), action_mask)


def compute_rewards(token_level_scores: torch.Tensor, log_probs: torch.Tensor, ref_log_probs: torch.Tensor, kl_ratio: float) -> torch.Tensor:
    """token-level scores minus kl_ratio x (log_probs - ref_log_probs) (core_algos.py:281-288)."""
    return token_level_scores - (log_probs - ref_log_probs) * kl_ratio


def compute_kl(log_probs: torch.Tensor, ref_log_probs: torch.Tensor, kl_penalty: str) -> torch.Tensor:
    """Per-token KL estimators on host tensors (core_algos.py:394-436) — the reward-side penalty branch and inspection; the
    training loss evaluates the same formulas inside st_grpo_loss."""
    lp, ref = log_probs.float(), ref_log_probs.float()
    if kl_penalty == "kl":
        return lp - ref
    if kl_penalty == "abs":
        return (lp - ref).abs()
    if kl_penalty == "mse":
        return 0.5 * (lp - ref).square()
    if kl_penalty == "low_var_kl":
        d = ref - lp
        return torch.clamp(d.exp() - d - 1, min=-10, max=10)
    if kl_penalty == "chi2":
        return torch.clamp(((ref - lp).exp() - 1) ** 2, min=0, max=20)
    if kl_penalty == "full":
        return torch.nn.functional.kl_div(ref, lp, log_target=True, reduction="none").sum(-1)
    raise NotImplementedError(f"Unknown KL penalty: {kl_penalty}.")
