"""The reference's tensor-level helpers under the reference's names (verl/utils/torch_functional.py).

Host bookkeeping on (batch, length) tensors — masked_mean, get_response_mask, pad_2d_list_to_length, postprocess_data — is plain
torch.  The heavy entries are the HIP library behind the reference's own signatures, so code written against the reference imports
and runs unchanged:
  log_probs_from_logits(logits, labels)   :34-66    -> autograd.Function over st_logprob_fwd / st_logprob_bwd
  AnyPrecisionAdamW(params, lr, ...)      :201-329  -> torch.optim.Optimizer whose step() is st_adamw_kahan_step per parameter
  get_constant_schedule_with_warmup(...)  :187-197  -> LambdaLR with the same multiplier
The training engine (spatialthinker_amd/actor.py) calls the same kernels on its flat buffers directly."""
from __future__ import annotations

from typing import Iterable, List, Optional, Tuple, Union

import torch
from torch.optim.lr_scheduler import LambdaLR


def masked_mean(values: torch.Tensor, mask: torch.Tensor, dim=None, eps: float = 1e-8) -> torch.Tensor:
    """:69-71."""
    return (values * mask).sum(dim=dim) / (mask.sum(dim=dim) + eps)


def masked_var(values: torch.Tensor, mask: torch.Tensor, unbiased: bool = True) -> torch.Tensor:
    """Variance over the masked entries, Bessel-corrected when `unbiased` and more than one entry is selected (torch_functional.py:74-88)."""
    mean = masked_mean(values, mask)
    var = masked_mean((values - mean) ** 2, mask)
    if unbiased:
        n = mask.sum()
        if n <= 1:
            print("The sum of the mask is less than one, which can cause a division by zero.")
            return var
        var = var * (n / (n - 1))
    return var


def masked_whiten(values: torch.Tensor, mask: torch.Tensor, eps: float = 1e-8) -> torch.Tensor:
    """(values - masked mean) * rsqrt(masked unbiased variance + eps) (torch_functional.py:91-94)."""
    return (values - masked_mean(values, mask)) * torch.rsqrt(masked_var(values, mask) + eps)


def pad_sequence_to_length(tensor: torch.Tensor, max_seq_len: int, pad_token_id: int, left_pad: bool = False) -> torch.Tensor:
    """Pad the LAST dim of an n-D tensor to max_seq_len with pad_token_id; longer tensors come back unchanged (torch_functional.py:137-147)."""
    short = max_seq_len - tensor.size(-1)
    if short <= 0:
        return tensor
    pad = torch.full((*tensor.shape[:-1], short), pad_token_id, dtype=tensor.dtype, device=tensor.device)
    return torch.cat((pad, tensor) if left_pad else (tensor, pad), dim=-1)


def get_response_mask(response_ids: torch.Tensor, eos_token_id: Union[int, List[int]] = 2, dtype: torch.dtype = torch.long) -> torch.Tensor:
    """:97-119 — 1 up to and INCLUDING the first EOS (any id of the list), 0 after it.
    e.g. eos = 1: ids [0, 0, 2, 4, 1, 5, 1] -> mask [1, 1, 1, 1, 1, 0, 0]."""
    ids = [eos_token_id] if isinstance(eos_token_id, int) else list(eos_token_id)
    is_eos = torch.zeros_like(response_ids, dtype=torch.bool)
    for e in ids:
        is_eos |= response_ids.eq(e)
    seen_before = torch.cumsum(is_eos.long(), dim=1) - is_eos.long()        # EOS tokens strictly before this position
    return seen_before.eq(0).to(dtype)


def pad_2d_list_to_length(response: List[List[int]], pad_token_id: int, max_length: Optional[int] = None) -> torch.Tensor:
    """:122-134 — right-pad ragged rows to max(max_length, longest row)."""
    longest = max(len(r) for r in response)
    target = max_length if (max_length is not None and max_length > longest) else longest
    return torch.tensor([list(r) + [pad_token_id] * (target - len(r)) for r in response])


def postprocess_data(input_ids, attention_mask, position_ids, max_length: int, pad_token_id: int, left_pad: bool = True,
                     truncation: str = "error"):
    """:150-184 — left-pad with pad/0/0 or truncate to max_length."""
    n = input_ids.shape[-1]
    if n < max_length:
        def pad(t, value):
            p = torch.full(t.shape[:-1] + (max_length - n,), value, dtype=t.dtype)
            return torch.cat((p, t), -1) if left_pad else torch.cat((t, p), -1)
        return pad(input_ids, pad_token_id), pad(attention_mask, 0), pad(position_ids, 0)
    if n > max_length:
        if truncation == "left":
            sl = slice(n - max_length, None)
        elif truncation == "right":
            sl = slice(0, max_length)
        else:
            raise NotImplementedError(f"{n} is larger than {max_length}.")
        return input_ids[..., sl], attention_mask[..., sl], position_ids[..., sl]
    return input_ids, attention_mask, position_ids


# ------------------------------------------------------------------ HIP-backed entries under the reference's names
class _LogProbsFromLogits(torch.autograd.Function):
    """logp[t] = logits[t, label[t]] - logsumexp(logits[t, :]) in one pass over the (T, V) bf16 logits (fp32 statistics); the
    backward overwrites the logits buffer with its gradient, as flash-attn's `cross_entropy_loss(..., inplace_backward=True)` does
    at verl/utils/torch_functional.py:26-31."""

    @staticmethod
    def forward(ctx, logits2d: torch.Tensor, labels1d: torch.Tensor):
        from spatialthinker_amd import ops
        logp, lse = ops.logprob_fwd(logits2d, labels1d, 1.0)
        ctx.save_for_backward(logits2d, labels1d, lse)
        return logp

    @staticmethod
    def backward(ctx, grad_out: torch.Tensor):
        from spatialthinker_amd import ops
        logits2d, labels1d, lse = ctx.saved_tensors
        ops.logprob_bwd_(logits2d, labels1d, lse, grad_out.contiguous().float(), 1.0)          # in place: logits <- dlogits
        return logits2d, None


def log_probs_from_logits(logits: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
    """:44-66 — log-probs of `labels` under `logits` (..., V) -> (...) fp32, the flash-attn sign convention (-NLL).
    bf16 logits on the GPU (what the actor produces, dp_actor.py:125-128); anything else is converted first."""
    batch_dim, vocab = logits.shape[:-1], logits.shape[-1]
    dev = logits.device if logits.is_cuda else torch.device("cuda", torch.cuda.current_device())
    z = logits.to(dev, torch.bfloat16).contiguous().view(-1, vocab)
    lab = labels.to(dev, torch.int64).contiguous().view(-1)
    return _LogProbsFromLogits.apply(z, lab).view(*batch_dim).to(logits.device)


def log_probs_from_logits_flash_attn(logits: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
    """The reference's name for the fused path (flash-attn's cross_entropy_loss, torch_functional.py:35-42); here both names reach the same
    HIP kernel."""
    return log_probs_from_logits(logits, labels)


def get_constant_schedule_with_warmup(optimizer: torch.optim.Optimizer, num_warmup_steps: int, last_epoch: int = -1):
    """:187-197 — lr * min(1, step / max(1, num_warmup_steps)); step 0 of a zero-warm-up schedule therefore trains at lr = 0
    (PolicyEngine.current_lr reproduces the same multiplier on its own counter)."""
    return LambdaLR(optimizer, lambda step: min(1.0, float(step) / float(max(1, num_warmup_steps))), last_epoch)


class AnyPrecisionAdamW(torch.optim.Optimizer):
    """:201-329 — AdamW with bf16 momentum / variance / Kahan-compensation buffers.  One fused HIP pass per parameter
    (st_adamw_kahan_step) with the rounding points of the reference's op sequence executed by torch on the GPU (bit-exact:
    tests/test_gpu_kernels.py).  Parameters must be bf16 CUDA tensors; gradients may be bf16 or fp32 (rounded to bf16 first, the
    dtype the reference's optimizer sees).  Only the configuration the reference constructs is built (fsdp_workers.py:292-299:
    Kahan summation on, bf16 states)."""

    def __init__(self, params: Iterable[torch.Tensor], lr: float = 1e-3, betas: Tuple[float, float] = (0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 0.0, use_kahan_summation: bool = True, momentum_dtype: torch.dtype = torch.bfloat16,
                 variance_dtype: torch.dtype = torch.bfloat16, compensation_buffer_dtype: torch.dtype = torch.bfloat16):
        if not use_kahan_summation or any(d != torch.bfloat16 for d in (momentum_dtype, variance_dtype, compensation_buffer_dtype)):
            raise NotImplementedError("AnyPrecisionAdamW: only use_kahan_summation=True with bf16 state dtypes is built (the reference's configuration)")
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, use_kahan_summation=use_kahan_summation,
                        momentum_dtype=momentum_dtype, variance_dtype=variance_dtype, compensation_buffer_dtype=compensation_buffer_dtype)
        super().__init__(params, defaults)

    @torch.no_grad()
    def step(self, closure=None):
        from spatialthinker_amd import ops
        if closure is not None:
            with torch.enable_grad():
                closure()
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.grad.is_sparse:
                    raise RuntimeError("AnyPrecisionAdamW does not support sparse gradients.")
                if p.dtype != torch.bfloat16 or not p.is_cuda or not p.is_contiguous():
                    raise TypeError("AnyPrecisionAdamW (HIP): parameters must be contiguous bf16 CUDA tensors")
                state = self.state[p]
                if len(state) == 0:
                    state["step"] = torch.tensor(0.0)
                    state["exp_avg"] = torch.zeros_like(p, dtype=torch.bfloat16)
                    state["exp_avg_sq"] = torch.zeros_like(p, dtype=torch.bfloat16)
                    state["compensation"] = torch.zeros_like(p, dtype=torch.bfloat16)
                state["step"] += 1
                g32 = p.grad.detach().contiguous().view(-1).float()
                ops.adamw_kahan_step_(p.data.view(-1), g32, state["exp_avg"].view(-1), state["exp_avg_sq"].view(-1),
                                      state["compensation"].view(-1), t=int(state["step"].item()), lr=group["lr"], betas=tuple(group["betas"]),
                                      eps=group["eps"], weight_decay=group["weight_decay"])
