"""Rollout post-processing: the (b*n, P+R) batch the trainer expects from `generate_sequences`
(reference: verl/workers/rollout/vllm_rollout_spmd.py:144-188).  The token generation itself is
spatialthinker_amd.rollout.Generator (prefill + hipGraph decode on the actor's weights); this is the integer bookkeeping
around it, kept separate so it is testable without a GPU (tests/test_trainer_math.py against the golden rl_extra.npz ro_*)."""
from __future__ import annotations

from typing import Dict, List, Union

import torch

from ...utils import torch_functional as VF


def assemble_rollout_batch(input_ids: torch.Tensor, attention_mask: torch.Tensor, position_ids: torch.Tensor, responses: torch.Tensor,
                           n: int, eos_token_id: Union[int, List[int]]) -> Dict[str, torch.Tensor]:
    """input_ids / attention_mask (b, P) left-padded prompts, position_ids (b, 3, P) or (b, P); responses (b*n, R) right-padded,
    prompt-major (rows [i*n, (i+1)*n) belong to prompt i).  Returns prompts, responses, input_ids, attention_mask, response_mask,
    position_ids with the reference's shapes: prompt tensors repeated n times (interleaved), response position ids continuing
    last+1 .. last+R on every M-RoPE row (also past the EOS), response mask = 1 through the first EOS."""
    if n > 1:
        input_ids, attention_mask, position_ids = (t.repeat_interleave(n, dim=0) for t in (input_ids, attention_mask, position_ids))
    if responses.shape[0] != input_ids.shape[0]:
        raise RuntimeError(f"{responses.shape[0]} responses for {input_ids.shape[0]} prompt rows")
    R = responses.shape[1]
    delta = torch.arange(1, R + 1, device=position_ids.device)
    delta = delta.view(1, 1, -1) if position_ids.dim() == 3 else delta.view(1, -1)
    resp_pos = position_ids[..., -1:] + delta
    resp_mask = VF.get_response_mask(responses, eos_token_id, dtype=attention_mask.dtype)
    return {"prompts": input_ids, "responses": responses, "input_ids": torch.cat([input_ids, responses], dim=-1),
            "attention_mask": torch.cat([attention_mask, resp_mask], dim=-1), "response_mask": resp_mask,
            "position_ids": torch.cat([position_ids, resp_pos], dim=-1)}
