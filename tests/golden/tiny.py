"""Deterministic tiny Qwen2.5-VL configuration + weights shared by the golden generator
(`make_golden.py`, runs in the build container against HF) and the tests (which run on the
GPU box where neither /root/reference nor network exist).  Weights come from numpy's legacy
Mersenne-Twister stream so they are bit-identical on every machine; nothing but outputs is
stored in the fixtures.
"""
from __future__ import annotations

import numpy as np

TINY = dict(
    hidden_size=256, intermediate_size=512, num_layers=2, num_heads=2, num_kv_heads=1,
    vocab_size=1024, rms_eps=1e-6, rope_theta=1e6, mrope_section=[16, 24, 24], tie_word_embeddings=False,
    v_depth=3, v_hidden=320, v_heads=4, v_intermediate=200, v_patch=14, v_temporal_patch=2, v_merge=2,
    v_window=56, v_fullatt=[1], v_in_channels=3, image_token_id=1010, vision_start_token_id=1011,
)
VISION_END, PAD_ID, EOS_ID = 1012, 1013, 1014


def param_shapes(c: dict) -> "dict[str, tuple]":
    H, I, V = c["hidden_size"], c["intermediate_size"], c["vocab_size"]
    hd = H // c["num_heads"]
    kv = c["num_kv_heads"] * hd
    vh, vi = c["v_hidden"], c["v_intermediate"]
    s = {"model.visual.patch_embed.proj.weight": (vh, c["v_in_channels"], c["v_temporal_patch"], c["v_patch"], c["v_patch"])}
    for i in range(c["v_depth"]):
        b = f"model.visual.blocks.{i}."
        s.update({b + "norm1.weight": (vh,), b + "norm2.weight": (vh,),
                  b + "attn.qkv.weight": (3 * vh, vh), b + "attn.qkv.bias": (3 * vh,),
                  b + "attn.proj.weight": (vh, vh), b + "attn.proj.bias": (vh,),
                  b + "mlp.gate_proj.weight": (vi, vh), b + "mlp.gate_proj.bias": (vi,),
                  b + "mlp.up_proj.weight": (vi, vh), b + "mlp.up_proj.bias": (vi,),
                  b + "mlp.down_proj.weight": (vh, vi), b + "mlp.down_proj.bias": (vh,)})
    m = c["v_merge"] ** 2 * vh
    s.update({"model.visual.merger.ln_q.weight": (vh,), "model.visual.merger.mlp.0.weight": (m, m),
              "model.visual.merger.mlp.0.bias": (m,), "model.visual.merger.mlp.2.weight": (H, m),
              "model.visual.merger.mlp.2.bias": (H,), "model.language_model.embed_tokens.weight": (V, H)})
    for i in range(c["num_layers"]):
        b = f"model.language_model.layers.{i}."
        s.update({b + "self_attn.q_proj.weight": (H, H), b + "self_attn.q_proj.bias": (H,),
                  b + "self_attn.k_proj.weight": (kv, H), b + "self_attn.k_proj.bias": (kv,),
                  b + "self_attn.v_proj.weight": (kv, H), b + "self_attn.v_proj.bias": (kv,),
                  b + "self_attn.o_proj.weight": (H, H), b + "mlp.gate_proj.weight": (I, H),
                  b + "mlp.up_proj.weight": (I, H), b + "mlp.down_proj.weight": (H, I),
                  b + "input_layernorm.weight": (H,), b + "post_attention_layernorm.weight": (H,)})
    s["model.language_model.norm.weight"] = (H,)
    if not c["tie_word_embeddings"]:
        s["lm_head.weight"] = (V, H)
    return s


def make_params(c: dict = TINY, seed: int = 1234, scale: float = 0.05, bf16_exact: bool = True) -> "dict[str, np.ndarray]":
    """N(0, scale) matrices, norm weights 1+N(0,0.1), biases N(0,0.02); optionally rounded so
    every value is exactly representable in bf16 (so fp32 oracle and bf16 engine see the same
    weights)."""
    rs = np.random.RandomState(seed)
    out = {}
    for name, shape in param_shapes(c).items():
        if name.endswith("norm.weight") or "norm1" in name or "norm2" in name or "ln_q" in name or "layernorm" in name:
            w = 1.0 + 0.1 * rs.standard_normal(shape)
        elif name.endswith(".bias"):
            w = 0.02 * rs.standard_normal(shape)
        else:
            w = scale * rs.standard_normal(shape)
        w = w.astype(np.float32)
        if bf16_exact:
            u = w.view(np.uint32).astype(np.uint64)
            w = (((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16).astype(np.uint32).view(np.float32)
        out[name] = w
    return out


def make_batch(c: dict = TINY, seed: int = 7, grids=((1, 8, 8), (1, 4, 12)), text_lens=((5, 9), (3, 6)),
               response_lens=(7, 11), R: int = 12, P: int = 64):
    """Two image+text samples in the reference's (B, P+R) layout: left-padded prompts, right-
    padded responses (verl/workers/rollout/vllm_rollout_spmd.py:177-188).  Returns numpy arrays:
    input_ids (B,S), attention_mask (B,S), responses (B,R), pixel_values (sumN, 1176),
    image_grid_thw (n,3) and per-sample patch counts."""
    rs = np.random.RandomState(seed)
    m2 = c["v_merge"] ** 2
    B = len(grids)
    ids = np.full((B, P + R), PAD_ID, dtype=np.int64)
    mask = np.zeros((B, P + R), dtype=np.int64)
    pix, counts = [], []
    for b, (g, (t0, t1), rl) in enumerate(zip(grids, text_lens, response_lens)):
        n_patch = g[0] * g[1] * g[2]
        n_img_tok = n_patch // m2
        prompt = (rs.randint(0, 900, size=t0).tolist() + [c["vision_start_token_id"]] + [c["image_token_id"]] * n_img_tok
                  + [VISION_END] + rs.randint(0, 900, size=t1).tolist())
        assert len(prompt) <= P
        ids[b, P - len(prompt):P] = prompt
        mask[b, P - len(prompt):P] = 1
        resp = rs.randint(0, 900, size=rl - 1).tolist() + [EOS_ID]
        ids[b, P:P + rl] = resp
        mask[b, P:P + rl] = 1
        pix.append(rs.standard_normal((n_patch, c["v_in_channels"] * c["v_temporal_patch"] * c["v_patch"] ** 2)).astype(np.float32))
        counts.append(n_patch)
    return dict(input_ids=ids, attention_mask=mask, responses=ids[:, P:].copy(),
                pixel_values=np.concatenate(pix, 0), image_grid_thw=np.asarray(grids, dtype=np.int64),
                patch_counts=np.asarray(counts), P=P, R=R)
