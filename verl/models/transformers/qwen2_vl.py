"""`get_rope_index(processor, input_ids, image_grid_thw, ..., attention_mask)` with the reference's signature
(verl/models/transformers/qwen2_vl.py:36-136), backed by the host-side index code of the MI355X build."""
from typing import Optional

import numpy as np
import torch

from spatialthinker_amd.indexing import get_rope_index as _rope_index


def get_rope_index(processor, input_ids: torch.Tensor, image_grid_thw: Optional[torch.Tensor] = None, video_grid_thw=None,
                   second_per_grid_ts=None, attention_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    if video_grid_thw is not None:
        raise NotImplementedError("video inputs are outside the SpatialThinker path")
    tok = processor.tokenizer
    out = _rope_index(input_ids.cpu().numpy(), None if image_grid_thw is None else image_grid_thw.cpu().numpy(),
                      None if attention_mask is None else attention_mask.cpu().numpy(),
                      image_token_id=tok.convert_tokens_to_ids("<|image_pad|>"), vision_start_token_id=tok.convert_tokens_to_ids("<|vision_start|>"),
                      spatial_merge_size=processor.image_processor.merge_size)
    return torch.from_numpy(out).to(input_ids.device)
