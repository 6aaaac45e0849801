"""Oracle (test infrastructure): Qwen2.5-VL forward in plain torch fp32 on CPU.

A restatement of the third-party model arithmetic the reference loads through
`AutoModelForVision2Seq.from_pretrained` (verl/workers/fsdp_workers.py:193-207) and calls
in `DataParallelPPOActor._forward_micro_batch` (verl/workers/actor/dp_actor.py:64-153,
padding-free path).  HF file = transformers/models/qwen2_5_vl/modeling_qwen2_5_vl.py
(5.15.0 in the build container; reference pin transformers>=4.49).

Parameters are a flat dict keyed by the HF state_dict names, so the golden fixtures can be
produced by HF itself and loaded here unchanged.  Everything is float32; autograd through
this function is the backward oracle.  Sequences are PACKED (T tokens, `cu_seqlens`), as on
the reference's padding-free path.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

from . import positions as P


@dataclass
class VLConfig:
    # text (HF Qwen2_5_VLTextConfig)
    hidden_size: int = 3584
    intermediate_size: int = 18944
    num_layers: int = 28
    num_heads: int = 28
    num_kv_heads: int = 4
    vocab_size: int = 152064
    rms_eps: float = 1e-6
    rope_theta: float = 1e6
    mrope_section: List[int] = field(default_factory=lambda: [16, 24, 24])
    tie_word_embeddings: bool = False
    # vision (HF Qwen2_5_VLVisionConfig)
    v_depth: int = 32
    v_hidden: int = 1280
    v_heads: int = 16
    v_intermediate: int = 3420
    v_patch: int = 14
    v_temporal_patch: int = 2
    v_merge: int = 2
    v_window: int = 112
    v_fullatt: List[int] = field(default_factory=lambda: [7, 15, 23, 31])
    v_in_channels: int = 3
    image_token_id: int = 151655
    vision_start_token_id: int = 151652

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_heads

    @property
    def v_head_dim(self) -> int:
        return self.v_hidden // self.v_heads


def rms_norm(x, w, eps):
    """HF :65-79: statistics in fp32, `.to(input_dtype)` before the weight.  For fp32 inputs (the oracle proper) both casts are the
    identity; with bf16 tensors on a GPU the same code is the "plain torch bf16 evaluation" yardstick of tests/test_gpu_depth.py."""
    xf = x.to(torch.float32)
    var = xf.pow(2).mean(-1, keepdim=True)
    return w * (xf * torch.rsqrt(var + eps)).to(x.dtype)


def rotate_half(x):
    """HF :153-157."""
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


def mrope_cos_sin(position_ids: torch.Tensor, head_dim: int, theta: float, section: List[int]):
    """HF :525-538 (rotary_emb.forward) + :557-599 (section select): position_ids (3,T) ->
    cos, sin (T, head_dim) with the frequency band i taking its angle from row
    [t,h,w][chunk(i) % 3], chunks = section*2 over the duplicated (freqs, freqs) layout."""
    inv_freq = (1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float32) / head_dim))).to(position_ids.device)
    freqs = position_ids.to(torch.float32)[:, :, None] * inv_freq[None, None, :]   # (3,T,hd/2)
    emb = torch.cat((freqs, freqs), dim=-1)                                        # (3,T,hd)
    cos, sin = emb.cos(), emb.sin()
    sel, parts_c, parts_s, off = section * 2, [], [], 0
    for i, n in enumerate(sel):
        parts_c.append(cos[i % 3, :, off:off + n])
        parts_s.append(sin[i % 3, :, off:off + n])
        off += n
    return torch.cat(parts_c, -1), torch.cat(parts_s, -1)


def dense_attention(q, k, v, cu_seqlens, causal: bool):
    """Per-segment softmax attention, fp32.  q (T,Hq,D), k/v (T,Hkv,D) -> (T,Hq,D).
    Equivalent of flash_attn_varlen_func as called at
    verl/models/transformers/flash_attention_utils.py:118-130 (causal) and HF :263-279 (ViT)."""
    T, Hq, D = q.shape
    rep = Hq // k.shape[1]
    out = torch.empty_like(q)
    scale = 1.0 / math.sqrt(D)
    bounds = [int(x) for x in cu_seqlens]
    for a, b in zip(bounds[:-1], bounds[1:]):
        if b == a:
            continue
        qs = q[a:b].transpose(0, 1)                                   # (Hq,L,D)
        ks = k[a:b].transpose(0, 1).repeat_interleave(rep, dim=0)     # HF repeat_kv :174-183
        vs = v[a:b].transpose(0, 1).repeat_interleave(rep, dim=0)
        s = torch.matmul(qs, ks.transpose(1, 2)) * scale
        if causal:
            L = b - a
            s = s.masked_fill(torch.ones(L, L, dtype=torch.bool, device=s.device).triu(1), float("-inf"))
        out[a:b] = torch.matmul(torch.softmax(s, dim=-1, dtype=torch.float32).to(vs.dtype), vs).transpose(0, 1)   # HF eager: fp32 softmax
    return out


def vision_tower(p: Dict[str, torch.Tensor], cfg: VLConfig, pixel_values: torch.Tensor, grid_thw,
                 taps: Optional[dict] = None) -> torch.Tensor:
    """HF :408-473 (`Qwen2_5_VisionTransformerPretrainedModel.forward`).
    pixel_values (N, C*Tp*14*14) -> merged image embeddings (N/4, out_hidden)."""
    pre = "model.visual."
    unit = cfg.v_merge ** 2
    N = pixel_values.shape[0]
    w_pe = p[pre + "patch_embed.proj.weight"].reshape(cfg.v_hidden, -1)           # Conv3d k=s => GEMM (:99-122)
    x = pixel_values.to(w_pe.dtype) @ w_pe.t()                                      # HF :118 casts the pixels to the weight dtype
    if taps is not None:
        taps["patch_embed"] = x.detach().clone()
    win_idx, cu_win = P.vision_window_index(grid_thw, merge_size=cfg.v_merge, window_size=cfg.v_window,
                                            patch_size=cfg.v_patch)
    win_idx_t = torch.from_numpy(win_idx)
    x = x.reshape(N // unit, unit, -1)[win_idx_t].reshape(N, -1)
    pos = torch.from_numpy(P.vision_position_ids(grid_thw, cfg.v_merge))           # (N,2)
    hd = cfg.v_head_dim
    inv_freq = 1.0 / (10000.0 ** (torch.arange(0, hd // 2, 2, dtype=torch.float32) / (hd // 2)))   # :125-135
    rot = (pos.to(torch.float32)[:, :, None] * inv_freq[None, None, :]).flatten(1)  # (N, hd/2)
    rot = rot.reshape(N // unit, unit, -1)[win_idx_t].reshape(N, -1)
    emb = torch.cat((rot, rot), dim=-1)
    cos, sin = emb.cos()[:, None, :].to(x.device, x.dtype), emb.sin()[:, None, :].to(x.device, x.dtype)
    cu_full = P.vision_cu_seqlens(grid_thw)
    for i in range(cfg.v_depth):
        x = vit_block(p, cfg, i, x, cos, sin, cu_full if i in cfg.v_fullatt else cu_win)
        if taps is not None:
            taps[f"vit_block{i}"] = x.detach().clone()
    m = pre + "merger."
    h = rms_norm(x, p[m + "ln_q.weight"], 1e-6).reshape(N // unit, -1)              # :137-150
    h = F.gelu(h @ p[m + "mlp.0.weight"].t() + p[m + "mlp.0.bias"])
    h = h @ p[m + "mlp.2.weight"].t() + p[m + "mlp.2.bias"]
    out = h[torch.argsort(win_idx_t)]
    if taps is not None:
        taps["image_embeds"] = out.detach().clone()
    return out


def vit_block(p, cfg: VLConfig, i: int, x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, cu) -> torch.Tensor:
    """One vision block (HF Qwen2_5_VLVisionBlock :286-322) on window-ordered patches x (N, v_hidden); cos/sin (N, 1, head_dim)."""
    b = f"model.visual.blocks.{i}."
    N, hd = x.shape[0], cfg.v_head_dim
    h = rms_norm(x, p[b + "norm1.weight"], 1e-6)
    qkv = h @ p[b + "attn.qkv.weight"].t() + p[b + "attn.qkv.bias"]
    q, k, v = qkv.reshape(N, 3, cfg.v_heads, hd).unbind(1)
    q = q * cos + rotate_half(q) * sin                                              # :160-171
    k = k * cos + rotate_half(k) * sin
    a = dense_attention(q, k, v, cu, causal=False).reshape(N, -1)
    x = x + a @ p[b + "attn.proj.weight"].t() + p[b + "attn.proj.bias"]
    h = rms_norm(x, p[b + "norm2.weight"], 1e-6)
    g = h @ p[b + "mlp.gate_proj.weight"].t() + p[b + "mlp.gate_proj.bias"]
    u = h @ p[b + "mlp.up_proj.weight"].t() + p[b + "mlp.up_proj.bias"]
    return x + (F.silu(g) * u) @ p[b + "mlp.down_proj.weight"].t() + p[b + "mlp.down_proj.bias"]


def forward_logits(p: Dict[str, torch.Tensor], cfg: VLConfig, input_ids: torch.Tensor,
                   position_ids: torch.Tensor, cu_seqlens, pixel_values: Optional[torch.Tensor] = None,
                   grid_thw=None, taps: Optional[dict] = None, rows: Optional[torch.Tensor] = None) -> torch.Tensor:
    """HF Qwen2_5_VLModel.forward :1185-1255 + TextModel :790-873 + lm_head, on packed input.

    input_ids (T,), position_ids (3,T), cu_seqlens (n+1,) -> logits (T,V) fp32 (or (len(rows),V)
    when `rows` selects token rows before the final norm / lm_head — both are row-wise)."""
    lm = "model.language_model."
    x = p[lm + "embed_tokens.weight"][input_ids]
    if pixel_values is not None:
        img = vision_tower(p, cfg, pixel_values, grid_thw, taps)
        mask = input_ids == cfg.image_token_id
        assert int(mask.sum()) == img.shape[0], "image tokens != image features"
        x = x.clone()
        x[mask] = img.to(x.dtype)                                                   # masked_scatter :1209-1215
    cos, sin = mrope_cos_sin(position_ids, cfg.head_dim, cfg.rope_theta, cfg.mrope_section)
    cos, sin = cos.to(x.dtype), sin.to(x.dtype)                                     # HF :536-538 (identity for the fp32 oracle)
    for i in range(cfg.num_layers):
        x = lm_layer(p, cfg, i, x, cos, sin, cu_seqlens)
        if taps is not None:
            taps[f"lm_layer{i}"] = x.detach().clone()
    if rows is not None:
        x = x[rows]
    return lm_head(p, cfg, x)


def lm_layer(p, cfg: VLConfig, i: int, x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, cu_seqlens, kv_out: Optional[list] = None):
    """One decoder layer (HF Qwen2_5_VLDecoderLayer :692-758) on packed rows x (T, H); cos/sin (T, head_dim).
    kv_out: if a list, the layer's roped K and V (T, Hkv, D) are appended (prefill of the KV-cache decode below)."""
    b = f"model.language_model.layers.{i}."
    T, D = x.shape[0], cfg.head_dim
    c, s_ = cos[:, None, :], sin[:, None, :]
    h = rms_norm(x, p[b + "input_layernorm.weight"], cfg.rms_eps)
    q = (h @ p[b + "self_attn.q_proj.weight"].t() + p[b + "self_attn.q_proj.bias"]).reshape(T, cfg.num_heads, D)
    k = (h @ p[b + "self_attn.k_proj.weight"].t() + p[b + "self_attn.k_proj.bias"]).reshape(T, cfg.num_kv_heads, D)
    v = (h @ p[b + "self_attn.v_proj.weight"].t() + p[b + "self_attn.v_proj.bias"]).reshape(T, cfg.num_kv_heads, D)
    q = q * c + rotate_half(q) * s_
    k = k * c + rotate_half(k) * s_
    if kv_out is not None:
        kv_out.append((k, v))
    a = dense_attention(q, k, v, cu_seqlens, causal=True).reshape(T, -1)
    x = x + a @ p[b + "self_attn.o_proj.weight"].t()
    h = rms_norm(x, p[b + "post_attention_layernorm.weight"], cfg.rms_eps)
    g = h @ p[b + "mlp.gate_proj.weight"].t()
    u = h @ p[b + "mlp.up_proj.weight"].t()
    return x + (F.silu(g) * u) @ p[b + "mlp.down_proj.weight"].t()


def lm_head(p, cfg: VLConfig, x: torch.Tensor) -> torch.Tensor:
    """final RMSNorm + lm_head (tied to the embedding table for the 3B model) on rows x (n, H) -> logits (n, V)."""
    lm = "model.language_model."
    x = rms_norm(x, p[lm + "norm.weight"], cfg.rms_eps)
    head = p[lm + "embed_tokens.weight"] if cfg.tie_word_embeddings else p["lm_head.weight"]
    return x @ head.t()


def lm_layer_decode(p, cfg: VLConfig, i: int, x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, k_cache: torch.Tensor,
                    v_cache: torch.Tensor, lens: torch.Tensor):
    """One decoder layer for ONE new token per sequence with a KV cache: x (B, H), cos/sin (B, head_dim), k_cache / v_cache
    (B, S_max, Hkv, D) holding lens[b] valid positions; the new K/V are written at position lens[b] (in place) and the token attends
    to positions [0, lens[b]].  The plain-PyTorch decode loop the CPU baseline times."""
    b_ = f"model.language_model.layers.{i}."
    B, D = x.shape[0], cfg.head_dim
    c, s_ = cos[:, None, :], sin[:, None, :]
    h = rms_norm(x, p[b_ + "input_layernorm.weight"], cfg.rms_eps)
    q = (h @ p[b_ + "self_attn.q_proj.weight"].t() + p[b_ + "self_attn.q_proj.bias"]).reshape(B, cfg.num_heads, D)
    k = (h @ p[b_ + "self_attn.k_proj.weight"].t() + p[b_ + "self_attn.k_proj.bias"]).reshape(B, cfg.num_kv_heads, D)
    v = (h @ p[b_ + "self_attn.v_proj.weight"].t() + p[b_ + "self_attn.v_proj.bias"]).reshape(B, cfg.num_kv_heads, D)
    q = q * c + rotate_half(q) * s_
    k = k * c + rotate_half(k) * s_
    ar = torch.arange(B, device=x.device)
    k_cache[ar, lens] = k
    v_cache[ar, lens] = v
    rep = cfg.num_heads // cfg.num_kv_heads
    S = int(lens.max()) + 1
    ks = k_cache[:, :S].repeat_interleave(rep, dim=2)                       # (B, S, Hq, D)
    vs = v_cache[:, :S].repeat_interleave(rep, dim=2)
    sc = torch.einsum("bhd,bshd->bhs", q, ks) / math.sqrt(D)
    sc = sc.masked_fill(torch.arange(S, device=x.device)[None, None, :] > lens[:, None, None], float("-inf"))
    a = torch.einsum("bhs,bshd->bhd", torch.softmax(sc, dim=-1), vs).reshape(B, -1)
    x = x + a @ p[b_ + "self_attn.o_proj.weight"].t()
    h = rms_norm(x, p[b_ + "post_attention_layernorm.weight"], cfg.rms_eps)
    g = h @ p[b_ + "mlp.gate_proj.weight"].t()
    u = h @ p[b_ + "mlp.up_proj.weight"].t()
    return x + (F.silu(g) * u) @ p[b_ + "mlp.down_proj.weight"].t()


@torch.no_grad()
def generate_greedy(p, cfg: VLConfig, input_ids: torch.Tensor, position_ids: torch.Tensor, max_new_tokens: int,
                    pixel_values: Optional[torch.Tensor] = None, grid_thw=None) -> torch.Tensor:
    """Greedy continuation of ONE un-padded prompt with a KV cache: prefill -> argmax -> lm_layer_decode loop.  position ids of
    the new tokens continue last+1.. on all three M-RoPE rows (vllm_rollout_spmd.py:159-170).  Returns (max_new_tokens,) ids."""
    lm = "model.language_model."
    T = input_ids.shape[0]
    x = p[lm + "embed_tokens.weight"][input_ids]
    if pixel_values is not None:
        img = vision_tower(p, cfg, pixel_values, grid_thw)
        x = x.clone()
        x[input_ids == cfg.image_token_id] = img.to(x.dtype)
    cos, sin = mrope_cos_sin(position_ids, cfg.head_dim, cfg.rope_theta, cfg.mrope_section)
    S_max = T + max_new_tokens
    kc = [torch.zeros(1, S_max, cfg.num_kv_heads, cfg.head_dim) for _ in range(cfg.num_layers)]
    vc = [torch.zeros(1, S_max, cfg.num_kv_heads, cfg.head_dim) for _ in range(cfg.num_layers)]
    for i in range(cfg.num_layers):
        kv: list = []
        x = lm_layer(p, cfg, i, x, cos, sin, [0, T], kv_out=kv)
        kc[i][0, :T], vc[i][0, :T] = kv[0]
    tok = int(lm_head(p, cfg, x[-1:]).argmax(-1))
    out = [tok]
    pos = position_ids[:, -1:] + 1
    lens = torch.tensor([T])
    for _ in range(max_new_tokens - 1):
        h = p[lm + "embed_tokens.weight"][torch.tensor([tok])]
        c1, s1 = mrope_cos_sin(pos, cfg.head_dim, cfg.rope_theta, cfg.mrope_section)
        for i in range(cfg.num_layers):
            h = lm_layer_decode(p, cfg, i, h, c1, s1, kc[i], vc[i], lens)
        tok = int(lm_head(p, cfg, h).argmax(-1))
        out.append(tok)
        pos, lens = pos + 1, lens + 1
    return torch.tensor(out)


def response_log_probs(p, cfg: VLConfig, input_ids_2d, attention_mask_2d, position_ids_3d, response_length: int,
                       temperature: float = 1.0, pixel_values=None, grid_thw=None):
    """verl/workers/actor/dp_actor.py:64-153, padding-free branch, restated:
    pack valid tokens -> model -> logits/temperature -> logp of the NEXT token (labels =
    roll(ids,-1), :104) -> scatter back to (B,S) -> slice [:, -R-1:-1].  Returns (B,R) fp32."""
    B, S = input_ids_2d.shape
    valid = attention_mask_2d.bool()
    flat_idx = valid.reshape(-1).nonzero()[:, 0]
    ids = input_ids_2d.reshape(-1)[flat_idx]
    pos = position_ids_3d.permute(1, 0, 2).reshape(3, -1)[:, flat_idx]            # (B,3,S)->(3,B*S)
    lens = valid.sum(-1)
    cu = torch.cat([lens.new_zeros(1), lens.cumsum(0)])
    logits = forward_logits(p, cfg, ids, pos, cu, pixel_values, grid_thw) / temperature
    labels = torch.roll(ids, -1)
    logp = torch.log_softmax(logits.to(torch.float32), dim=-1).gather(-1, labels[:, None])[:, 0]
    full = torch.zeros(B * S, dtype=logp.dtype, device=logp.device)
    full[flat_idx] = logp
    return full.reshape(B, S)[:, -response_length - 1:-1]


def response_values(p, cfg: VLConfig, input_ids_2d, attention_mask_2d, position_ids_3d, response_length: int, pixel_values=None, grid_thw=None):
    """verl/workers/critic/dp_critic.py:52-125 restated (padding-free branch): the token-classification model's `logits` — score =
    Linear(H, 1) (p["score.weight"] (1, H), p["score.bias"] (1,)) on the final-norm hidden state of every valid token — scattered back to
    (B, S) and sliced [:, -R-1:-1]: the value of the state BEFORE each response token.  Returns (B, R) fp32."""
    B, S = input_ids_2d.shape
    valid = attention_mask_2d.bool()
    flat_idx = valid.reshape(-1).nonzero()[:, 0]
    ids = input_ids_2d.reshape(-1)[flat_idx]
    pos = position_ids_3d.permute(1, 0, 2).reshape(3, -1)[:, flat_idx]
    lens = valid.sum(-1)
    cu = torch.cat([lens.new_zeros(1), lens.cumsum(0)])
    lm = "model.language_model."
    x = p[lm + "embed_tokens.weight"][ids]
    if pixel_values is not None:
        img = vision_tower(p, cfg, pixel_values, grid_thw)
        mask = ids == cfg.image_token_id
        x = x.clone()
        x[mask] = img.to(x.dtype)
    cos, sin = mrope_cos_sin(pos, cfg.head_dim, cfg.rope_theta, cfg.mrope_section)
    cos, sin = cos.to(x.dtype), sin.to(x.dtype)
    for i in range(cfg.num_layers):
        x = lm_layer(p, cfg, i, x, cos, sin, cu)
    h = rms_norm(x, p[lm + "norm.weight"], cfg.rms_eps)
    v = (h @ p["score.weight"].reshape(-1, 1))[:, 0] + p["score.bias"].reshape(-1)[0]
    full = torch.zeros(B * S, dtype=v.dtype, device=v.device)
    full[flat_idx] = v
    return full.reshape(B, S)[:, -response_length - 1:-1]

