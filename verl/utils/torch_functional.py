"""Host tensor helpers with the reference's names (verl/utils/torch_functional.py).  The heavy entries of that file live in the
HIP library (log_probs_from_logits -> st_logprob_fwd/bwd, AnyPrecisionAdamW -> st_adamw_kahan_step); what remains here is the
integer / bookkeeping work the workers and the trainer do on (batch, length) host tensors."""
from __future__ import annotations

from typing import List, Optional, Union

import torch


def masked_mean(values: torch.Tensor, mask: torch.Tensor, dim=None, eps: float = 1e-8) -> torch.Tensor:
    """:69-71."""
    return (values * mask).sum(dim=dim) / (mask.sum(dim=dim) + eps)


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
