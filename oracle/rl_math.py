"""Oracle (test infrastructure): GRPO / PPO scalar math, log-probs and the bf16 AdamW.

numpy float32/float64 restatements.  Reference citations are relative to
/root/reference (hunarbatra/SpatialThinker).
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np
import torch


# --------------------------------------------------------------------------- helpers
def bf16_round(x: np.ndarray) -> np.ndarray:
    """Round-to-nearest-even fp32 -> bf16 -> fp32, done on the bit pattern."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    u = x.view(np.uint32).astype(np.uint64)
    nan = np.isnan(x)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    out = r.astype(np.uint32).view(np.float32).copy()
    out[nan] = np.nan
    return out


def masked_mean(values, mask, eps: float = 1e-8):
    """verl/utils/torch_functional.py:69-71 — sum(v*m) / (sum(m) + eps), fp32."""
    values = np.asarray(values, dtype=np.float32)
    mask = np.asarray(mask, dtype=np.float32)
    return np.float32((values * mask).sum(dtype=np.float32) / (mask.sum(dtype=np.float32) + np.float32(eps)))


def response_mask(response_ids: np.ndarray, eos_token_id) -> np.ndarray:
    """verl/utils/torch_functional.py:97-119 — 1 up to and including the first EOS."""
    ids = np.asarray(response_ids)
    eos = [eos_token_id] if isinstance(eos_token_id, int) else list(eos_token_id)
    is_eos = np.zeros(ids.shape, dtype=bool)
    for e in eos:
        is_eos |= ids == e
    seen_before = (np.cumsum(is_eos, axis=1) - is_eos.astype(np.int64)) > 0
    return (~seen_before).astype(np.int64)


# --------------------------------------------------------------------------- log-probs
def log_probs_from_logits(logits: np.ndarray, labels: np.ndarray) -> np.ndarray:
    """verl/utils/torch_functional.py:34-42 (flash-attn CE semantics, i.e. -NLL):
    logp[t] = logits[t, label] - logsumexp(logits[t, :]), fp32 math on the given logits.
    (The torch fallback at :63-64 returns +NLL — SURVEY.md §0.7 — not followed.)"""
    z = np.asarray(logits, dtype=np.float32)
    m = z.max(axis=-1, keepdims=True)
    lse = m[..., 0] + np.log(np.exp((z - m).astype(np.float64)).sum(axis=-1)).astype(np.float32)
    picked = np.take_along_axis(z, np.asarray(labels)[..., None], axis=-1)[..., 0]
    return (picked - lse).astype(np.float32)


def log_probs_grad(logits: np.ndarray, labels: np.ndarray, g: np.ndarray) -> np.ndarray:
    """d(sum_t g[t]*logp[t]) / dlogits = g[t] * (onehot(label) - softmax(logits[t]))."""
    z = np.asarray(logits, dtype=np.float64)
    m = z.max(axis=-1, keepdims=True)
    p = np.exp(z - m)
    p /= p.sum(axis=-1, keepdims=True)
    d = -p
    np.put_along_axis(d, np.asarray(labels)[..., None], np.take_along_axis(d, np.asarray(labels)[..., None], -1) + 1.0, -1)
    return (d * np.asarray(g, dtype=np.float64)[..., None]).astype(np.float32)


# --------------------------------------------------------------------------- GRPO
def grpo_outcome_advantage(token_level_rewards, resp_mask, index, eps: float = 1e-6):
    """verl/trainer/core_algos.py:137-175.

    score_i = sum_t reward[i,t]; per uid group (in first-appearance order):
    mean and UNBIASED std (torch.std, N-1) in fp32; A_i = (score_i-mean)/(std+eps);
    advantages = returns = A_i * response_mask.
    """
    r = np.asarray(token_level_rewards, dtype=np.float32)
    scores = r.sum(axis=-1, dtype=np.float32)
    groups: "OrderedDict[object, list[int]]" = OrderedDict()
    for i, uid in enumerate(index):
        groups.setdefault(uid, []).append(i)
    out = np.zeros_like(scores)
    for uid, rows in groups.items():
        assert len(rows) > 1, "GRPO needs rollout.n > 1."
        vals = torch.tensor([float(scores[j]) for j in rows], dtype=torch.float32)
        mean = vals.mean()
        std = vals.std()  # unbiased
        for j in rows:
            out[j] = ((torch.tensor(scores[j]) - mean) / (std + eps)).item()
    adv = out[:, None] * np.asarray(resp_mask, dtype=np.float32)
    return adv.astype(np.float32), adv.astype(np.float32)


def policy_loss(old_log_probs, log_probs, advantages, resp_mask,
                clip_ratio_low: float, clip_ratio_high: float, clip_ratio_dual: float):
    """verl/trainer/core_algos.py:291-353 (dual-clip PPO), fp32.

    Returns (pg_loss, clipfrac_higher, clipfrac_lower, ppo_kl) as np.float32 scalars."""
    old = np.asarray(old_log_probs, np.float32)
    new = np.asarray(log_probs, np.float32)
    adv = np.asarray(advantages, np.float32)
    d = new - old
    ratio = np.exp(d)
    lo, hi = np.float32(np.log(1.0 - clip_ratio_low)), np.float32(np.log(1.0 + clip_ratio_high))
    clipped = np.exp(np.clip(d, lo, hi))
    l1 = -adv * ratio
    l2 = -adv * clipped
    l3 = -adv * np.float32(clip_ratio_dual)
    higher = np.maximum(l1, l2)
    frac_hi = (l1 < l2).astype(np.float32)
    lower = np.minimum(higher, l3)
    final = np.where(adv < 0, lower, higher)
    frac_lo = (higher > l3).astype(np.float32) * (adv < 0).astype(np.float32)
    return (masked_mean(final, resp_mask), masked_mean(frac_hi, resp_mask),
            masked_mean(frac_lo, resp_mask), masked_mean(-d, resp_mask))


def policy_loss_grad(old_log_probs, log_probs, advantages, resp_mask,
                     clip_ratio_low, clip_ratio_high, clip_ratio_dual):
    """d pg_loss / d log_probs, derived from core_algos.py:331-349 (autograd semantics of
    torch.max/min/where/clamp: gradient flows to the selected branch; clamp passes gradient
    only strictly inside... torch.clamp passes it on the closed interval [lo, hi])."""
    old = np.asarray(old_log_probs, np.float32)
    new = np.asarray(log_probs, np.float32)
    adv = np.asarray(advantages, np.float32)
    m = np.asarray(resp_mask, np.float32)
    d = new - old
    lo, hi = np.float32(np.log(1.0 - clip_ratio_low)), np.float32(np.log(1.0 + clip_ratio_high))
    ratio = np.exp(d)
    clipped = np.exp(np.clip(d, lo, hi))
    l1, l2, l3 = -adv * ratio, -adv * clipped, -adv * np.float32(clip_ratio_dual)
    g1 = -adv * ratio                                   # d l1 / d new
    inside = ((d >= lo) & (d <= hi)).astype(np.float32)
    g2 = -adv * clipped * inside                         # d l2 / d new
    # torch.max(a, b): ties split the gradient 0.5/0.5
    sel1 = np.where(l1 > l2, 1.0, np.where(l1 == l2, 0.5, 0.0)).astype(np.float32)
    g_hi = sel1 * g1 + (1 - sel1) * g2
    higher = np.maximum(l1, l2)
    selh = np.where(higher < l3, 1.0, np.where(higher == l3, 0.5, 0.0)).astype(np.float32)
    g_lo = selh * g_hi                                   # l3 has no dependence on new
    g = np.where(adv < 0, g_lo, g_hi)
    return (g * m / (m.sum(dtype=np.float32) + np.float32(1e-8))).astype(np.float32)


def kl_penalty(log_probs, ref_log_probs, kind: str):
    """verl/trainer/core_algos.py:394-436 (kl / abs / mse / low_var_kl / chi2), fp32."""
    lp = np.asarray(log_probs, np.float32)
    ref = np.asarray(ref_log_probs, np.float32)
    if kind == "kl":
        return lp - ref
    if kind == "abs":
        return np.abs(lp - ref)
    if kind == "mse":
        return np.float32(0.5) * np.square(lp - ref)
    if kind == "low_var_kl":
        d = ref - lp
        return np.clip(np.exp(d) - d - np.float32(1), np.float32(-10), np.float32(10))
    if kind == "chi2":
        r = np.exp(ref - lp)
        return np.clip(np.square(r - np.float32(1)), np.float32(0), np.float32(20))
    raise NotImplementedError(kind)


def kl_penalty_grad(log_probs, ref_log_probs, kind: str):
    """d kl_penalty / d log_probs (elementwise), clamp gradient on the closed interval."""
    lp = np.asarray(log_probs, np.float32)
    ref = np.asarray(ref_log_probs, np.float32)
    if kind == "kl":
        return np.ones_like(lp)
    if kind == "abs":
        return np.sign(lp - ref).astype(np.float32)
    if kind == "mse":
        return lp - ref
    if kind == "low_var_kl":
        d = ref - lp
        raw = np.exp(d) - d - np.float32(1)
        inside = ((raw >= -10) & (raw <= 10)).astype(np.float32)
        return (-(np.exp(d)) + np.float32(1)) * inside
    if kind == "chi2":
        r = np.exp(ref - lp)
        raw = np.square(r - np.float32(1))
        inside = ((raw >= 0) & (raw <= 20)).astype(np.float32)
        return (np.float32(2) * (r - np.float32(1)) * (-r)) * inside
    raise NotImplementedError(kind)


def actor_micro_batch_loss(log_probs, old_log_probs, ref_log_probs, advantages, resp_mask, *,
                           clip_ratio_low=0.2, clip_ratio_high=0.3, clip_ratio_dual=3.0,
                           kl_kind="low_var_kl", kl_coef=1e-2, grad_accum=1):
    """verl/workers/actor/dp_actor.py:252-278 — one micro-batch of update_policy.

    Returns dict(metrics) and g = d(loss/grad_accum)/d log_probs."""
    pg, fh, fl, ppo_kl = policy_loss(old_log_probs, log_probs, advantages, resp_mask,
                                     clip_ratio_low, clip_ratio_high, clip_ratio_dual)
    g = policy_loss_grad(old_log_probs, log_probs, advantages, resp_mask,
                         clip_ratio_low, clip_ratio_high, clip_ratio_dual)
    out = {"entropy_loss": -masked_mean(log_probs, resp_mask), "pg_clipfrac_higher": fh,
           "pg_clipfrac_lower": fl, "ppo_kl": ppo_kl}
    if ref_log_probs is not None:
        kld = kl_penalty(log_probs, ref_log_probs, kl_kind)
        kl_loss = masked_mean(kld, resp_mask)
        m = np.asarray(resp_mask, np.float32)
        g = g + np.float32(kl_coef) * kl_penalty_grad(log_probs, ref_log_probs, kl_kind) * m / (
            m.sum(dtype=np.float32) + np.float32(1e-8))
        pg = np.float32(pg + kl_loss * np.float32(kl_coef))
        out["kl_loss"] = kl_loss
    out["pg_loss"] = pg
    return out, (g / np.float32(grad_accum)).astype(np.float32)


# --------------------------------------------------------------------------- optimizer
def _fma32(a, b, c):
    """float32 fused multiply-add (single rounding), emulated through float64."""
    return (np.asarray(a, np.float64) * np.asarray(b, np.float64) + np.asarray(c, np.float64)).astype(np.float32)


class AdamWKahanBF16:
    """verl/utils/torch_functional.py:253-329 (AnyPrecisionAdamW.step) with bf16 param,
    bf16 grad, bf16 exp_avg / exp_avg_sq / compensation.  Every torch op on a bf16 tensor
    computes in fp32 and rounds once to bf16; this restatement spells each rounding out.

      p  = bf16(p * (1 - lr*wd))                                       (:300-301)
      m  = bf16(fma(A, g, bf16(m*b1))),   A = 1-b1                     (:303)
      v  = bf16(fma((1-b2)*g, g, bf16(v*b2)))                          (:304)  [fused multiply-add:
                                                   probed bit-exact against torch's addcmul_]
      bc1 = 1 - b1**t (fp32); step_size = lr/bc1; dc = sqrt(1-b2**t)   (:306-309)
      cv = bf16(bf16(bf16(sqrt(v)) / dc) + E),  E = eps                (:310)
      c  = bf16(c + ((-step_size) * m) / cv)                           (:314)
      p' = bf16(p + c);  c = bf16(c + bf16(p - p'))                    (:318-320)

    `scalar_mode` selects how the python scalars of `add_(g, alpha=A)` and `.add_(E)` enter
    the arithmetic — the one place where torch's CPU and GPU elementwise kernels differ
    (probed with torch 2.10): on CPU they are first cast to the tensor dtype (A = bf16(1-b1)
    = 0.10009765625, E = bf16(1e-8)); on the GPU (CUDA and ROCm alike — the only device the
    reference can run on, fsdp_workers.py:75-76) they stay fp32 "opmath" scalars.
      * "cpu": pinned BIT-EXACT against tests/golden/adamw.npz (reference class run here);
      * "gpu": what the HIP kernel implements; cross-checked on the GPU box against torch-ROCm's
        own bf16 elementwise ops in tests/test_gpu_kernels.py.
    """

    def __init__(self, lr=1e-6, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, scalar_mode="gpu"):
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        assert scalar_mode in ("cpu", "gpu")
        self.scalar_mode = scalar_mode
        self.t = 0
        self.m = self.v = self.c = None

    def scalars(self, t: int, lr: float):
        """The step-dependent scalars exactly as torch computes them: `step` is a float32 0-d
        tensor, so beta**step, 1-x, lr/x and x**0.5 are all float32 ops."""
        b1, b2 = self.betas
        step = torch.tensor(float(t), dtype=torch.float32)
        bc1 = 1 - b1 ** step
        step_size = lr / bc1
        dc = (1 - b2 ** step) ** 0.5
        return float(step_size), float(dc)

    def step(self, p: np.ndarray, g: np.ndarray, lr: float | None = None) -> np.ndarray:
        lr = self.lr if lr is None else lr
        b1, b2 = self.betas
        p = bf16_round(p)
        g = bf16_round(g)
        if self.m is None:
            self.m, self.v, self.c = np.zeros_like(p), np.zeros_like(p), np.zeros_like(p)
        self.t += 1
        step_size, dc = self.scalars(self.t, lr)
        f = np.float32
        A, E = f(1 - b1), f(self.eps)
        if self.scalar_mode == "cpu":
            A, E = bf16_round(np.array([A]))[0], bf16_round(np.array([E]))[0]
        if self.wd:
            p = bf16_round(p * f(1 - lr * self.wd))
        self.m = bf16_round(_fma32(A, g, bf16_round(self.m * f(b1))))
        if self.scalar_mode == "cpu":      # CPU addcmul: fma(value*t1, t2, self)
            self.v = bf16_round(_fma32(f(1 - b2) * g, g, bf16_round(self.v * f(b2))))
        else:                              # GPU addcmul: fma(value, t1*t2, self)   (probed on MI355X, tools/probe_adamw.py)
            self.v = bf16_round(_fma32(f(1 - b2), g * g, bf16_round(self.v * f(b2))))
        cv = bf16_round(bf16_round(bf16_round(np.sqrt(self.v)) / f(dc)) + E)
        if self.scalar_mode == "cpu":      # CPU kernel: self + (value*t1)/t2
            self.c = bf16_round(self.c + (f(-step_size) * self.m) / cv)
        else:                              # GPU kernel: fma(value, t1/t2, self)
            self.c = bf16_round(_fma32(f(-step_size), self.m / cv, self.c))
        p_new = bf16_round(p + self.c)
        self.c = bf16_round(self.c + bf16_round(p - p_new))
        return p_new


def constant_schedule_lr(base_lr: float, num_warmup_steps: int, scheduler_steps_taken: int) -> float:
    """verl/utils/torch_functional.py:187-197: lr_lambda(s) = min(1, s/max(1, warmup)); note
    lambda(0) = 0, so the first update_actor call trains at lr 0."""
    return base_lr * min(1.0, float(scheduler_steps_taken) / float(max(1, num_warmup_steps)))
