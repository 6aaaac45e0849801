"""perf/mfu_actor with the reference's estimate (verl/utils/flops_counter.py:82-115: 6*N_dense*tokens + 12*sum(s^2)*head_dim*heads*
layers, LM only, embedding counted as a GEMM) and an MI355X entry the reference's device table lacks (:40-55)."""
from typing import List, Tuple

MI355X_BF16_DENSE_FLOPS = 2.5e15


def get_device_flops(unit: str = "T") -> float:
    """Dense bf16 peak of the device in `unit` flop/s (B, K, M, G, T, P): the reference's table (flops_counter.py:27-55: H100 / A100 /
    L40 / L20 / H20 / 910B, inf for an unknown device) plus the AMD parts it lacks (MI355X 2.5e15, MI300X 1.307e15 dense)."""
    import torch
    name = torch.cuda.get_device_name() if torch.cuda.is_available() else ""
    table = (("MI355", MI355X_BF16_DENSE_FLOPS), ("MI350", 2.3e15), ("MI325", 1.307e15), ("MI300X", 1.307e15), ("H100", 989e12), ("H800", 989e12),
             ("A100", 312e12), ("A800", 312e12), ("L40", 181.05e12), ("L20", 119.5e12), ("H20", 148e12), ("910B", 354e12))
    flops = next((v for k, v in table if k in name), MI355X_BF16_DENSE_FLOPS if not name else float("inf"))
    steps = ["B", "K", "M", "G", "T", "P"]
    if unit not in steps:
        raise ValueError(f"unit must be one of {steps}")
    return flops / (1000.0 ** steps.index(unit)) if flops > 0 else flops


class FlopsCounter:
    def __init__(self, config):
        self.cfg = config       # spatialthinker_amd.model.VLConfig (the reference takes the HF PretrainedConfig, flops_counter.py:66-80)

    def estimate_flops(self, batch_seqlens: List[int], delta_time: float) -> Tuple[float, float]:
        c = self.cfg
        hd = c.head_dim
        dense = (c.hidden_size * c.qkv_width + c.num_heads * hd * c.hidden_size + 3 * c.hidden_size * c.intermediate_size) * c.num_layers \
            + 2 * c.vocab_size * c.hidden_size
        tokens = sum(batch_seqlens)
        attn = 12 * sum(s * s for s in batch_seqlens) * hd * c.num_heads * c.num_layers
        return (6 * dense * tokens + attn) / delta_time / 1e12, MI355X_BF16_DENSE_FLOPS / 1e12
