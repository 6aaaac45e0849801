"""`get_rope_index(processor, input_ids, image_grid_thw, ..., attention_mask)` with the reference's signature
(verl/models/transformers/qwen2_vl.py:36-136), backed by the host-side index code of the MI355X build."""
from typing import Optional

import numpy as np
import torch

from spatialthinker_amd.indexing import get_rope_index as _rope_index


def get_rope_index(processor, input_ids: torch.Tensor, image_grid_thw: Optional[torch.Tensor] = None, video_grid_thw=None,
                   second_per_grid_ts=None, attention_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    tok = processor.tokenizer
    np_ = lambda t: None if t is None else (t.cpu().numpy() if torch.is_tensor(t) else np.asarray(t))
    out = _rope_index(input_ids.cpu().numpy(), np_(image_grid_thw), np_(attention_mask),
                      image_token_id=tok.convert_tokens_to_ids("<|image_pad|>"), vision_start_token_id=tok.convert_tokens_to_ids("<|vision_start|>"),
                      spatial_merge_size=processor.image_processor.merge_size, video_grid_thw=np_(video_grid_thw),
                      second_per_grid_ts=np_(second_per_grid_ts), video_token_id=tok.convert_tokens_to_ids("<|video_pad|>"))
    return torch.from_numpy(out).to(input_ids.device)
