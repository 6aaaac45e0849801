"""Real Qwen2.5-VL-7B / 3B dimensions on depth-reduced models (1 LM layer, 2 ViT blocks) + cheap deterministic bf16-exact weights,
shared by the GPU parity tests that exercise the production dispatch paths (tests/test_gpu_fullsize.py,
tests/test_gpu_production_shapes.py).  Pure numpy: importable on the GPU box (no reference, no network)."""
from __future__ import annotations

import numpy as np

import tiny

FULL = dict(hidden_size=3584, intermediate_size=18944, num_layers=1, num_heads=28, num_kv_heads=4, vocab_size=152064,
            rms_eps=1e-6, rope_theta=1e6, mrope_section=[16, 24, 24], tie_word_embeddings=False,
            v_depth=2, v_hidden=1280, v_heads=16, v_intermediate=3420, v_patch=14, v_temporal_patch=2, v_merge=2, v_window=112,
            v_fullatt=[1], v_in_channels=3, image_token_id=151655, vision_start_token_id=151652)
VISION_END, EOS, PAD = 151653, 151645, 151643
FULL_3B = dict(FULL, hidden_size=2048, intermediate_size=11008, num_heads=16, num_kv_heads=2, vocab_size=151936, tie_word_embeddings=True)


def bf16_round(w: np.ndarray) -> np.ndarray:
    """fp32 array rounded (RNE) to bf16-representable values, numpy only."""
    w = np.ascontiguousarray(w, dtype=np.float32)
    u = w.view(np.uint32).astype(np.uint64)
    return (((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16).astype(np.uint32).view(np.float32)


def make_params(seed: int = 3, dims: dict = None) -> "dict[str, np.ndarray]":
    """Cheap deterministic bf16-exact weights (the big tables from a ramp: value quality is irrelevant, cost is not)."""
    rs = np.random.RandomState(seed)
    out = {}
    for name, shape in tiny.param_shapes(dims or FULL).items():
        n = int(np.prod(shape))
        if "norm" in name or "ln_q" in name:
            w = 1.0 + 0.1 * rs.standard_normal(shape)
        elif name.endswith(".bias"):
            w = 0.02 * rs.standard_normal(shape)
        elif n > (1 << 24):
            w = (((np.arange(n, dtype=np.int64) * 2654435761) % 2039).astype(np.float32) / 2039.0 - 0.5).reshape(shape) * 0.04
        else:
            w = 0.02 * rs.standard_normal(shape)
        out[name] = bf16_round(w)
    return out
