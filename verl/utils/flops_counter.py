"""perf/mfu_actor with the reference's estimate (verl/utils/flops_counter.py:82-115: 6*N_dense*tokens + 12*sum(s^2)*head_dim*heads*
layers, LM only, embedding counted as a GEMM) and an MI355X entry the reference's device table lacks (:40-55)."""
from typing import List, Tuple

MI355X_BF16_DENSE_FLOPS = 2.5e15


class FlopsCounter:
    def __init__(self, cfg):
        self.cfg = cfg          # spatialthinker_amd.model.VLConfig

    def estimate_flops(self, batch_seqlens: List[int], delta_time: float) -> Tuple[float, float]:
        c = self.cfg
        hd = c.head_dim
        dense = (c.hidden_size * c.qkv_width + c.num_heads * hd * c.hidden_size + 3 * c.hidden_size * c.intermediate_size) * c.num_layers \
            + 2 * c.vocab_size * c.hidden_size
        tokens = sum(batch_seqlens)
        attn = 12 * sum(s * s for s in batch_seqlens) * hd * c.num_heads * c.num_layers
        return (6 * dense * tokens + attn) / delta_time / 1e12, MI355X_BF16_DENSE_FLOPS / 1e12
