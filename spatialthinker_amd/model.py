"""Qwen2.5-VL on MI355X: explicit forward / backward over flat parameter buffers.

No autograd, no tracing compiler: every layer is a fixed sequence of launches of the hand-written
HIP kernels in libst_hip.so (through `ops`), with the activations a backward needs kept in HBM
(288 GB per GPU makes per-layer recomputation — the reference's gradient checkpointing,
verl/workers/fsdp_workers.py:220-221 — unnecessary: training costs 3x a forward instead of 4x).

What this replaces in the reference: the HF `Qwen2_5_VLForConditionalGeneration` module loaded at
verl/workers/fsdp_workers.py:193-207 and called at verl/workers/actor/dp_actor.py:118-124, plus the
attention monkey-patch verl/models/transformers/qwen2_vl.py:139-189.

HBM layout
  * all weights live in ONE flat bf16 buffer (`ParamStore.flat`); gradients are a flat fp32 buffer with
    identical offsets (one fused AdamW launch, bucketed all-reduce = slices of one tensor);
  * q/k/v and gate/up projections are stored fused ([q|k|v] rows, [gate|up] rows) so each is one GEMM;
  * the ViT MLP width 3420 is zero-padded to 3456 and the patch-embed K 1176 to 1216 (multiples of 64 for
    the MFMA K loop); the pads stay exactly zero under training (zero gradient, decay of zero);
  * every weight used in a backward dX GEMM also has a transposed copy (refreshed after each optimizer
    step) so that all GEMMs are the K-contiguous "NT" form the kernel implements.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import os

import numpy as np
import torch

from . import indexing as ix
from . import ops

BF16, F32, I32, I64 = torch.bfloat16, torch.float32, torch.int32, torch.int64


@dataclass
class VLConfig:
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
    value_head: bool = False            # critic: score = Linear(H, 1) on the final hidden state instead of the lm_head (token classification)
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
    def head_dim(self):
        return self.hidden_size // self.num_heads

    @property
    def v_head_dim(self):
        return self.v_hidden // self.v_heads

    @property
    def qkv_width(self):
        return (self.num_heads + 2 * self.num_kv_heads) * self.head_dim

    @property
    def v_inter_pad(self):
        return ix.round_up(self.v_intermediate, 64)

    @property
    def patch_k(self):
        return self.v_in_channels * self.v_temporal_patch * self.v_patch * self.v_patch

    @property
    def patch_k_pad(self):
        return ix.round_up(self.patch_k, 64)

    @staticmethod
    def qwen2_5_vl_7b():
        return VLConfig()

    @staticmethod
    def qwen2_5_vl_3b():
        return VLConfig(hidden_size=2048, intermediate_size=11008, num_layers=36, num_heads=16, num_kv_heads=2,
                        vocab_size=151936, tie_word_embeddings=True)

    @staticmethod
    def from_hf_dict(d: dict) -> "VLConfig":
        t = d.get("text_config", d)
        v = d["vision_config"]
        rope = t.get("rope_parameters") or t.get("rope_scaling") or {}
        return VLConfig(hidden_size=t["hidden_size"], intermediate_size=t["intermediate_size"], num_layers=t["num_hidden_layers"],
                        num_heads=t["num_attention_heads"], num_kv_heads=t["num_key_value_heads"], vocab_size=t["vocab_size"],
                        rms_eps=t.get("rms_norm_eps", 1e-6), rope_theta=rope.get("rope_theta", t.get("rope_theta", 1e6)),
                        mrope_section=list(rope.get("mrope_section", [16, 24, 24])),
                        tie_word_embeddings=bool(d.get("tie_word_embeddings", t.get("tie_word_embeddings", False))),
                        v_depth=v["depth"], v_hidden=v["hidden_size"], v_heads=v["num_heads"], v_intermediate=v["intermediate_size"],
                        v_patch=v.get("patch_size", 14), v_temporal_patch=v.get("temporal_patch_size", 2),
                        v_merge=v.get("spatial_merge_size", 2), v_window=v.get("window_size", 112),
                        v_fullatt=list(v.get("fullatt_block_indexes", [7, 15, 23, 31])), v_in_channels=v.get("in_channels", 3),
                        image_token_id=d.get("image_token_id", 151655), vision_start_token_id=d.get("vision_start_token_id", 151652))

    def n_params(self) -> int:
        return sum(int(np.prod(s)) for s in param_layout(self).values())

    def flops_forward(self, seqlens: List[int], patches_per_image: List[int], logit_rows: Optional[int] = None) -> float:
        """Algorithmic forward FLOPs (matmul 2MNK, causal attention at half) — SURVEY.md §8(d), ViT included."""
        H, I, D = self.hidden_size, self.intermediate_size, self.head_dim
        T = sum(seqlens)
        per_tok = 2 * (H * self.qkv_width + H * H + 3 * H * I)
        fl = per_tok * T * self.num_layers
        fl += sum(2 * D * self.num_heads * s * s for s in seqlens) * self.num_layers            # QK^T + PV, causal half
        rows = T if logit_rows is None else logit_rows
        fl += 2 * H * self.vocab_size * rows
        vh, vi, hd = self.v_hidden, self.v_intermediate, self.v_head_dim
        N = sum(patches_per_image)
        fl += N * self.v_depth * 2 * (vh * 3 * vh + vh * vh + 3 * vh * vi) + N * 2 * self.patch_k * vh
        fl += (N // 4) * 2 * (4 * vh * 4 * vh + 4 * vh * H)
        n_full = len(self.v_fullatt)
        fl += sum(4 * hd * self.v_heads * (n_full * n * n + (self.v_depth - n_full) * n * 64) for n in patches_per_image)
        return float(fl)


    def flops_forward_grouped(self, groups: List[Tuple[int, List[int]]], patches_per_image: List[int], logit_rows: int,
                              prefix_cached: bool = False) -> float:
        """Forward FLOPs actually executed with shared-prompt packing: groups = [(prompt_len, [response_len, ...]), ...] — the
        prompt is computed once per group; a response row attends to the whole prompt plus its own causal part.
        prefix_cached: the prompt rows themselves are not computed at all (their K/V come from the rollout prefill)."""
        H, I, D = self.hidden_size, self.intermediate_size, self.head_dim
        T = sum((0 if prefix_cached else P) + sum(rs) for P, rs in groups)
        fl = 2 * (H * self.qkv_width + H * H + 3 * H * I) * T * self.num_layers
        att = sum((0 if prefix_cached else P * P) + sum(2 * r * P + r * r for r in rs) for P, rs in groups)   # (q, k) pairs x 2
        fl += 2 * D * self.num_heads * att * self.num_layers
        fl += 2 * H * self.vocab_size * logit_rows
        return float(fl) + self.flops_forward([], patches_per_image, logit_rows=0)



# ------------------------------------------------------------------------------------------
def param_layout(c: VLConfig) -> "Dict[str, Tuple[int, ...]]":
    """name -> shape of the engine's own (fused / padded) parameter tensors, in flat-buffer order."""
    H, I, V = c.hidden_size, c.intermediate_size, c.vocab_size
    vh, vi = c.v_hidden, c.v_inter_pad
    m = c.v_merge ** 2 * vh
    L: Dict[str, Tuple[int, ...]] = {"v.patch_embed": (vh, c.patch_k_pad)}
    for i in range(c.v_depth):
        p = f"v.{i}."
        L.update({p + "norm1": (vh,), p + "qkv_w": (3 * vh, vh), p + "qkv_b": (3 * vh,), p + "proj_w": (vh, vh), p + "proj_b": (vh,),
                  p + "norm2": (vh,), p + "gu_w": (2 * vi, vh), p + "gu_b": (2 * vi,), p + "down_w": (vh, vi), p + "down_b": (vh,)})
    L.update({"v.merger.ln_q": (vh,), "v.merger.fc1_w": (m, m), "v.merger.fc1_b": (m,), "v.merger.fc2_w": (H, m), "v.merger.fc2_b": (H,)})
    L["embed"] = (V, H)
    for i in range(c.num_layers):
        p = f"l.{i}."
        L.update({p + "in_norm": (H,), p + "qkv_w": (c.qkv_width, H), p + "qkv_b": (c.qkv_width,), p + "o_w": (H, H),
                  p + "post_norm": (H,), p + "gu_w": (2 * I, H), p + "down_w": (H, I)})
    L["final_norm"] = (H,)
    if c.value_head:
        L["score_w"] = (H,)
        L["score_b"] = (8,)                 # element 0 is the bias (8: keeps the slice a whole 16-byte vector)
    elif not c.tie_word_embeddings:
        L["lm_head"] = (V, H)
    return L


class ParamStore:
    """Flat bf16 weights (+ optional fp32 grads, bf16 AdamW states, transposed copies)."""

    def __init__(self, cfg: VLConfig, device="cuda", trainable: bool = False):
        self.cfg, self.device, self.trainable = cfg, device, trainable
        self.layout = param_layout(cfg)
        self.offsets: Dict[str, int] = {}
        off = 0
        for name, shape in self.layout.items():
            self.offsets[name] = off
            off += ix.round_up(int(np.prod(shape)), 64)          # 128-byte aligned slices
        self.numel = off
        self.flat = torch.zeros(off, dtype=BF16, device=device)
        self.w = {n: self._view(self.flat, n) for n in self.layout}
        self.grad = self.g = None
        self.m = self.v = self.c = None
        self.wq = None                   # MX-fp8 copies of the LM projection weights (refresh_fp8), only in fp8 mode
        self.wqt = None                  # ... transposed copies for the fp8 input-gradient GEMMs (enable_fp8(dgrad=True))
        self.master = None               # fp32 master weights (enable_fp32_master), None = the bf16 weights ARE the parameters
        if trainable:
            self.grad = torch.zeros(off, dtype=F32, device=device)
            self.g = {n: self._view(self.grad, n) for n in self.layout}
            self.m = torch.zeros(off, dtype=BF16, device=device)
            self.v = torch.zeros(off, dtype=BF16, device=device)
            self.c = torch.zeros(off, dtype=BF16, device=device)

    def enable_fp32_master(self):
        """The reference's default actor (worker.actor.fsdp.torch_dtype unset, verl/workers/fsdp_workers.py:186-189): fp32 parameters and
        fp32 AdamW moments; `flat` stays the bf16 working copy every kernel computes with (= FSDP's param_dtype=bf16 all-gather) and is
        re-rounded from the master by st_adamw_master_step.  12 bytes per parameter of state instead of 6 (bf16 m / v / Kahan)."""
        assert self.trainable
        # NOTE: built from the bf16 working copy; a caller that wants the master to carry a checkpoint's own fp32 values enables the
        # master FIRST and loads afterwards (load_hf_state_dict then fills the master from the fp32 tensors and rounds `flat` from
        # it) — pretrained.load_model(master_fp32=True) and FSDPWorker.init_model do
        self.master = self.flat.float()
        self.m = torch.zeros(self.numel, dtype=F32, device=self.device)
        self.v = torch.zeros(self.numel, dtype=F32, device=self.device)
        self.c = None

    def _view(self, flat, name):
        shape = self.layout[name]
        o = self.offsets[name]
        return flat[o:o + int(np.prod(shape))].view(*shape)

    def refresh_transposes(self):
        """Copies derived from the weights, rebuilt after every optimizer step / load.  There are no transposed copies any more: the
        backward reads every weight AS STORED (dX = dY W through st_gemm_nn, contraction-major B); only the fp8 mode keeps copies."""
        if getattr(self, "wq", None):
            self.refresh_fp8()

    def refresh_fp8(self):
        """MX-fp8 copies (e4m3 bytes + e8m0 block scales) of the LM projection weights for the fp8 forward GEMMs (Qwen25VL.fp8);
        re-quantised from the bf16 master weights after every optimizer step.  ~0.53 bytes per parameter on top of the bf16 store.
        fp8_dgrad: a second copy of each weight TRANSPOSED and quantised along the output dimension — dX = dY W contracts over it, and
        MX blocks must run along the contraction — for the fp8 input-gradient GEMMs (same size again)."""
        self.wq = {}
        self.wqt = {} if getattr(self, "fp8_dgrad", False) else None
        for i in range(self.cfg.num_layers):
            for nm in ("qkv_w", "o_w", "gu_w", "down_w"):
                w = self.w[f"l.{i}.{nm}"]
                self.wq[f"l.{i}.{nm}"] = ops.mxfp8_quantize(w)
                if self.wqt is not None:
                    self.wqt[f"l.{i}.{nm}"] = ops.mxfp8_quantize(ops.transpose(w))

    # ---- HF state_dict interop (names of transformers Qwen2_5_VLForConditionalGeneration) -------------
    def load_hf_state_dict(self, sd: Dict[str, torch.Tensor], target: Optional[torch.Tensor] = None):
        """HF-named tensors into the fused / padded flat layout.  target: another flat buffer with the same layout (AdamW moments,
        Kahan compensation: the reference's optimizer state is keyed by the same parameter names) instead of the weights."""
        c = self.cfg
        views = self.w if target is None else {n: self._view(target, n) for n in self.layout}

        def put(name, t):
            dst = views[name]
            t = t.to(dtype=dst.dtype, device=self.device)
            if dst.shape == t.shape:
                dst.copy_(t)
            elif dst.dim() == 2:
                dst.zero_()
                dst[:t.shape[0], :t.shape[1]].copy_(t)
            else:
                dst.zero_()
                dst[:t.shape[0]].copy_(t)

        vi, vip = c.v_intermediate, c.v_inter_pad
        put("v.patch_embed", sd["model.visual.patch_embed.proj.weight"].reshape(c.v_hidden, -1))
        for i in range(c.v_depth):
            s, p = f"model.visual.blocks.{i}.", f"v.{i}."
            put(p + "norm1", sd[s + "norm1.weight"]); put(p + "norm2", sd[s + "norm2.weight"])
            put(p + "qkv_w", sd[s + "attn.qkv.weight"]); put(p + "qkv_b", sd[s + "attn.qkv.bias"])
            put(p + "proj_w", sd[s + "attn.proj.weight"]); put(p + "proj_b", sd[s + "attn.proj.bias"])
            gu = torch.zeros(2 * vip, c.v_hidden); gb = torch.zeros(2 * vip)
            gu[:vi] = sd[s + "mlp.gate_proj.weight"].float(); gu[vip:vip + vi] = sd[s + "mlp.up_proj.weight"].float()
            gb[:vi] = sd[s + "mlp.gate_proj.bias"].float(); gb[vip:vip + vi] = sd[s + "mlp.up_proj.bias"].float()
            put(p + "gu_w", gu); put(p + "gu_b", gb)
            put(p + "down_w", sd[s + "mlp.down_proj.weight"]); put(p + "down_b", sd[s + "mlp.down_proj.bias"])
        put("v.merger.ln_q", sd["model.visual.merger.ln_q.weight"])
        put("v.merger.fc1_w", sd["model.visual.merger.mlp.0.weight"]); put("v.merger.fc1_b", sd["model.visual.merger.mlp.0.bias"])
        put("v.merger.fc2_w", sd["model.visual.merger.mlp.2.weight"]); put("v.merger.fc2_b", sd["model.visual.merger.mlp.2.bias"])
        put("embed", sd["model.language_model.embed_tokens.weight"])
        for i in range(c.num_layers):
            s, p = f"model.language_model.layers.{i}.", f"l.{i}."
            put(p + "in_norm", sd[s + "input_layernorm.weight"]); put(p + "post_norm", sd[s + "post_attention_layernorm.weight"])
            put(p + "qkv_w", torch.cat([sd[s + f"self_attn.{x}_proj.weight"].float() for x in "qkv"], 0))
            put(p + "qkv_b", torch.cat([sd[s + f"self_attn.{x}_proj.bias"].float() for x in "qkv"], 0))
            put(p + "o_w", sd[s + "self_attn.o_proj.weight"])
            put(p + "gu_w", torch.cat([sd[s + "mlp.gate_proj.weight"].float(), sd[s + "mlp.up_proj.weight"].float()], 0))
            put(p + "down_w", sd[s + "mlp.down_proj.weight"])
        put("final_norm", sd["model.language_model.norm.weight"])
        if c.value_head:                                        # a policy checkpoint has no score head: it starts at zero (values = 0)
            if "score.weight" in sd:
                put("score_w", sd["score.weight"].reshape(-1))
                if "score.bias" in sd:
                    bias = torch.zeros(8); bias[0] = float(sd["score.bias"].reshape(-1)[0]); put("score_b", bias)
        elif not c.tie_word_embeddings:
            put("lm_head", sd["lm_head.weight"])
        if self.trainable and target is None:
            if getattr(self, "master", None) is not None:        # fp32 master mode: the master takes the checkpoint's own precision
                self.load_hf_state_dict(sd, target=self.master)
            self.refresh_transposes()

    def export_hf(self, source: Optional[Dict[str, torch.Tensor]] = None) -> Dict[str, torch.Tensor]:
        """Un-fuse / un-pad `source` (default: the weights; pass self.g for gradients) back to HF names."""
        c = self.cfg
        src = self.w if source is None else source
        H, D = c.hidden_size, c.head_dim
        nq, nkv = c.num_heads * D, c.num_kv_heads * D
        vi, vip, I = c.v_intermediate, c.v_inter_pad, c.intermediate_size
        out = {"model.visual.patch_embed.proj.weight": src["v.patch_embed"][:, :c.patch_k].reshape(
            c.v_hidden, c.v_in_channels, c.v_temporal_patch, c.v_patch, c.v_patch)}
        for i in range(c.v_depth):
            s, p = f"model.visual.blocks.{i}.", f"v.{i}."
            out.update({s + "norm1.weight": src[p + "norm1"], s + "norm2.weight": src[p + "norm2"],
                        s + "attn.qkv.weight": src[p + "qkv_w"], s + "attn.qkv.bias": src[p + "qkv_b"],
                        s + "attn.proj.weight": src[p + "proj_w"], s + "attn.proj.bias": src[p + "proj_b"],
                        s + "mlp.gate_proj.weight": src[p + "gu_w"][:vi], s + "mlp.up_proj.weight": src[p + "gu_w"][vip:vip + vi],
                        s + "mlp.gate_proj.bias": src[p + "gu_b"][:vi], s + "mlp.up_proj.bias": src[p + "gu_b"][vip:vip + vi],
                        s + "mlp.down_proj.weight": src[p + "down_w"][:, :vi], s + "mlp.down_proj.bias": src[p + "down_b"]})
        out.update({"model.visual.merger.ln_q.weight": src["v.merger.ln_q"], "model.visual.merger.mlp.0.weight": src["v.merger.fc1_w"],
                    "model.visual.merger.mlp.0.bias": src["v.merger.fc1_b"], "model.visual.merger.mlp.2.weight": src["v.merger.fc2_w"],
                    "model.visual.merger.mlp.2.bias": src["v.merger.fc2_b"], "model.language_model.embed_tokens.weight": src["embed"]})
        for i in range(c.num_layers):
            s, p = f"model.language_model.layers.{i}.", f"l.{i}."
            qw, qb = src[p + "qkv_w"], src[p + "qkv_b"]
            out.update({s + "input_layernorm.weight": src[p + "in_norm"], s + "post_attention_layernorm.weight": src[p + "post_norm"],
                        s + "self_attn.q_proj.weight": qw[:nq], s + "self_attn.k_proj.weight": qw[nq:nq + nkv],
                        s + "self_attn.v_proj.weight": qw[nq + nkv:], s + "self_attn.q_proj.bias": qb[:nq],
                        s + "self_attn.k_proj.bias": qb[nq:nq + nkv], s + "self_attn.v_proj.bias": qb[nq + nkv:],
                        s + "self_attn.o_proj.weight": src[p + "o_w"], s + "mlp.gate_proj.weight": src[p + "gu_w"][:I],
                        s + "mlp.up_proj.weight": src[p + "gu_w"][I:], s + "mlp.down_proj.weight": src[p + "down_w"]})
        out["model.language_model.norm.weight"] = src["final_norm"]
        if c.value_head:
            out["score.weight"] = src["score_w"].reshape(1, -1)
            out["score.bias"] = src["score_b"][:1]
        elif not c.tie_word_embeddings:
            out["lm_head.weight"] = src["lm_head"]
        return out

    def init_random(self, seed: int = 0, std: float = 0.02):
        """normal(0, std) matrices, unit norm weights, zero biases (synthetic benchmarks: SURVEY.md §8d)."""
        g = torch.Generator(device=self.device).manual_seed(seed)
        c = self.cfg
        for n, t in self.w.items():
            if n.endswith(("norm", "norm1", "norm2", "ln_q")):
                t.fill_(1.0)
            elif n.endswith("_b"):
                t.zero_()
            else:
                t.copy_((torch.randn(t.shape, generator=g, device=self.device, dtype=F32) * std).to(BF16))
        vi, vip = c.v_intermediate, c.v_inter_pad
        if vip != vi:
            for i in range(c.v_depth):
                self.w[f"v.{i}.gu_w"][vi:vip].zero_(); self.w[f"v.{i}.gu_w"][vip + vi:].zero_()
                self.w[f"v.{i}.down_w"][:, vi:].zero_()
        if c.patch_k_pad != c.patch_k:
            self.w["v.patch_embed"][:, c.patch_k:].zero_()
        if self.trainable:
            self.refresh_transposes()


# ------------------------------------------------------------------------------------------
@dataclass
class DeviceBatch:
    """A packed micro-batch resident in HBM (built from indexing.PackedBatch + VisionPlan)."""
    pk: ix.PackedBatch
    ids: torch.Tensor
    embed_ids: torch.Tensor
    cu: torch.Tensor
    cos: torch.Tensor
    sin: torch.Tensor
    image_rows: torch.Tensor
    logit_rows: torch.Tensor
    labels: torch.Tensor
    out_index: torch.Tensor
    Tr_pad: int
    vis: Optional[dict] = None
    seg: Optional[tuple] = None          # (seg_b, seg_e, pre_b, pre_e, dep_e) int32 device arrays (shared-prefix attention)
    logit_dup: Optional[torch.Tensor] = None
    logit_distinct: Optional[torch.Tensor] = None
    first_prompt: Optional[torch.Tensor] = None
    pairs: float = 0.0                   # (query, key) pairs of one LM attention launch over this batch (profiling hooks only)


class Qwen25VL:
    def __init__(self, cfg: VLConfig, params: ParamStore):
        self.cfg, self.p = cfg, params
        self.sp_group, self.sp, self.sp_rank = None, 1, 0          # Ulysses sequence parallelism: set_sequence_parallel
        D = cfg.head_dim
        self.inv_freq = (1.0 / (cfg.rope_theta ** (torch.arange(0, D, 2, dtype=F32) / D))).to(params.device)   # HF :523
        self.scale = D ** -0.5
        self.v_scale = cfg.v_head_dim ** -0.5
        self.recompute_light = False     # see _lm_layer_fwd
        self.unfused_swiglu_with_grad = os.environ.get("ST_SWIGLU_UNFUSED_GRAD", "1") != "0"      # see _lm_layer_fwd
        self.fp8 = False                 # config #5: the LM's four projection GEMMs run forward in MX-fp8 (enable_fp8)
        self.fp8_dgrad = False           # ... and their input-gradient GEMMs too
        self.fp8_wgrad = False           # ... and their weight-gradient GEMMs

    # ---------------------------------------------------------------- Ulysses sequence parallelism
    def set_sequence_parallel(self, group) -> None:
        """Cut every packed pass into `sp` equal row slices over the ranks of `group` (reference verl/utils/ulysses.py + the attention
        patch flash_attention_utils.py:98-106,146-148): all row-wise work of the LM layers — norms, projections, MLP — runs on this rank's
        slice; around the attention kernel one all-to-all hands every rank ALL rows of n_heads / sp heads (the kernels see whole
        sequences and the unchanged segment tables) and a second one trades the result back.  None switches it off.
        First version: the embedding / ViT in front of the layers and the lm_head + log-prob behind them run on the full stream on
        every rank (replicated); only the rank's own logit rows enter its backward, so every weight gradient is a partial sum over the
        rank's tokens and the sp ranks' gradients ADD UP (actor.GradReducer: sum over the world, mean over world / sp)."""
        import torch.distributed as dist
        self.sp_group = group
        self.sp = dist.get_world_size(group) if group is not None else 1
        self.sp_rank = dist.get_rank(group) if group is not None else 0
        c = self.cfg
        if self.sp > 1 and (c.num_heads % self.sp or c.num_kv_heads % self.sp):
            raise ValueError(f"ulysses_sequence_parallel_size = {self.sp} must divide the {c.num_heads} query heads and the {c.num_kv_heads} key/value heads "
                             f"(this engine does not repeat K/V heads for sp > num_kv_heads as the reference's attention patch does: "
                             f"sp <= {c.num_kv_heads} for this model; INTEGRATION.md)")

    def _sp_a2a(self, x3: torch.Tensor, scatter_dim: int, gather_dim: int) -> torch.Tensor:
        """One all-to-all over the sequence-parallel group: `sp` equal pieces of x3 along scatter_dim, piece j to rank j, the received
        pieces concatenated in rank order along gather_dim."""
        import torch.distributed as dist
        send = torch.stack([p_.contiguous() for p_ in torch.tensor_split(x3, self.sp, dim=scatter_dim)], 0)
        recv = torch.empty_like(send)
        if dist.get_backend(self.sp_group) == "gloo":          # test mode (ranks sharing one GPU): gloo's all-to-all takes host tensors
            host = torch.empty(send.shape, dtype=send.dtype)
            dist.all_to_all_single(host, send.cpu(), group=self.sp_group)
            recv.copy_(host)
        else:
            dist.all_to_all_single(recv, send, group=self.sp_group)
        return torch.cat(list(recv.unbind(0)), dim=gather_dim).contiguous()

    def _sp_gather_seq(self, x: torch.Tensor, heads: int) -> torch.Tensor:
        """(T / sp, heads * D) rows of this rank -> (T, heads / sp * D): all rows, this rank's heads."""
        D = self.cfg.head_dim
        y = self._sp_a2a(x.reshape(x.shape[0], heads, D), 1, 0)
        return y.reshape(y.shape[0], (heads // self.sp) * D)

    def _sp_scatter_seq(self, x: torch.Tensor, heads: int) -> torch.Tensor:
        """(T, heads / sp * D) -> (T / sp, heads * D): the inverse trade."""
        D = self.cfg.head_dim
        y = self._sp_a2a(x.reshape(x.shape[0], heads // self.sp, D), 0, 1)
        return y.reshape(y.shape[0], heads * D)

    def _sp_rows(self, T_pad: int):
        if T_pad % self.sp:
            raise ValueError(f"packed length {T_pad} is not divisible by the sequence-parallel size {self.sp}")
        Tl = T_pad // self.sp
        return self.sp_rank * Tl, Tl

    def _sp_all_gather_rows(self, x: torch.Tensor) -> torch.Tensor:
        import torch.distributed as dist
        out = x.new_empty((x.shape[0] * self.sp,) + tuple(x.shape[1:]))
        if dist.get_backend(self.sp_group) == "gloo":
            host = torch.empty(out.shape, dtype=out.dtype)
            dist.all_gather_into_tensor(host, x.cpu().contiguous(), group=self.sp_group)
            out.copy_(host)
        else:
            dist.all_gather_into_tensor(out, x.contiguous(), group=self.sp_group)
        return out

    def enable_fp8(self, on: bool = True, dgrad: bool = False, wgrad: bool = False):
        """Forward GEMMs of the LM layers (qkv, o, gate/up, down — 93 % of the forward FLOPs) on the block-scaled fp8 MFMA path:
        activations are quantised on the fly (one HBM pass each), weights once per optimizer step.  Attention, norms, the lm_head,
        the ViT and the weight gradients stay bf16 (straight-through: gradients are those of the bf16 layer evaluated at the fp8
        forward's activations).  dgrad=True also runs the four input-gradient GEMMs dX = dY W of every LM layer in MX-fp8 (dY quantised
        on the fly along its feature dimension, a transposed fp8 copy of W); wgrad=True the weight-gradient GEMMs dW = dY^T X as well
        (both operands quantised token-minor on the fly).  Tolerance: DESIGN.md §4 (fp8)."""
        if on and (self.cfg.hidden_size % 128 or self.cfg.intermediate_size % 128):
            raise ValueError("fp8 mode needs hidden and intermediate sizes that are multiples of 128 (MX K-tiles)")
        self.fp8 = bool(on)
        self.fp8_dgrad = bool(on and dgrad)
        self.fp8_wgrad = bool(on and wgrad)
        self.p.fp8_dgrad = self.fp8_dgrad
        if on and (not getattr(self.p, "wq", None) or (self.fp8_dgrad and not getattr(self.p, "wqt", None))):
            self.p.refresh_fp8()
        if not on:
            self.p.wq = None
            self.p.wqt = None

    def _dgrad(self, dy, name, dyq=None):
        """dX = dY W for an LM projection: bf16 (W read as stored, contraction-major) or MX-fp8 against the transposed fp8 copy
        (dyq: dY's row-wise MX-fp8 operand when _quantize_grad already made it)."""
        if self.fp8_dgrad and dy.shape[0] > 256 and dy.shape[1] % 128 == 0:
            dq, ds = dyq if dyq is not None else ops.mxfp8_quantize(dy)
            wq, ws = self.p.wqt[name]
            return ops.gemm_mxfp8_nt(dq, ds, wq, ws)
        return ops.gemm_nn(dy, self.p.w[name])

    def _quantize_grad(self, dy):
        """A projection's output gradient feeds its input-gradient GEMM (MX blocks along the features) and its weight-gradient GEMM (blocks
        along the tokens): with both in fp8 the two operands come out of ONE pass over dY.  Returns (row-wise operand, token-minor operand)
        or (None, None) when the shapes / modes do not call for it."""
        if self.fp8_dgrad and self.fp8_wgrad and dy.shape[0] > 256 and dy.shape[0] % 128 == 0 and dy.shape[1] % 128 == 0:
            return ops.mxfp8_quantize_both(dy)
        return None, None

    def _linear(self, x, name, bias=None, residual=None, xq=None):
        """y = x W^T (+bias)(+residual) for an LM projection: bf16 MFMA GEMM, or MX-fp8 when enabled and the shape fits its tile
        (xq: the input's MX-fp8 operand (q, scales) when its producer already emitted it)."""
        if xq is not None:
            wq, ws = self.p.wq[name]
            return ops.gemm_mxfp8_nt(xq[0], xq[1], wq, ws, bias=bias, residual=residual)
        w = self.p.w[name]
        if self.fp8 and x.shape[0] > 256 and x.shape[1] % 128 == 0:
            xq, xs = ops.mxfp8_quantize(x)
            wq, ws = self.p.wq[name]
            return ops.gemm_mxfp8_nt(xq, xs, wq, ws, bias=bias, residual=residual)
        return ops.gemm_nt(x, w, bias=bias, residual=residual)

    # ---------------------------------------------------------------- batch staging
    def stage(self, input_ids, attention_mask, position_ids, response_length: int, pixel_values=None, image_grid_thw=None,
              image_map=None, groups=None) -> DeviceBatch:
        """image_map (optional): pixel_values / image_grid_thw hold each DISTINCT image once and image_map[j] names the image of the
        j-th image-bearing sample (one image per sample); the features are gathered per sample and their gradients summed."""
        c, dev = self.cfg, self.p.device
        pk = ix.pack_batch(_np(input_ids), _np(attention_mask), _np(position_ids), response_length, image_token_id=c.image_token_id,
                           groups=groups, value_rows=c.value_head)
        t = lambda a, dt: ops.h2d(a, dt, dev)
        pos = t(pk.pos, I32)
        cos, sin = ops.mrope_table(pos, self.inv_freq, c.head_dim, c.mrope_section)
        Tr = len(pk.logit_rows)
        Tr_pad = max(ix.round_up(Tr, 128), 128)
        rows = np.zeros(Tr_pad, dtype=np.int32); rows[:Tr] = pk.logit_rows
        labels = np.full(Tr_pad, -1, dtype=np.int64); labels[:Tr] = pk.labels
        vis = None
        if pixel_values is not None and len(pk.image_rows):
            plan = ix.plan_vision(_np(image_grid_thw), merge=c.v_merge, window=c.v_window, patch=c.v_patch, head_dim=c.v_head_dim)
            n_feat = plan.n_patches // (c.v_merge ** 2)
            img_src = dup_idx = None
            if image_map is not None:
                g_ = _np(image_grid_thw).reshape(-1, 3)
                n_tok = (g_[:, 0] * g_[:, 1] * g_[:, 2]) // (c.v_merge ** 2)
                off = np.concatenate([[0], np.cumsum(n_tok)])
                img_src = np.concatenate([np.arange(off[u], off[u + 1]) for u in image_map]).astype(np.int32)
                users = [[] for _ in range(len(n_tok))]
                pos0 = 0
                for u in image_map:                              # positions (in image_rows order) of every user of image u
                    users[u].append(pos0); pos0 += int(n_tok[u])
                kmax = max(len(x_) for x_ in users)
                dup_idx = -np.ones((n_feat, kmax), dtype=np.int32)
                for u, starts in enumerate(users):
                    for r_, st_ in enumerate(starts):
                        dup_idx[off[u]:off[u + 1], r_] = st_ + np.arange(int(n_tok[u]))
                assert len(img_src) == len(pk.image_rows), "image tokens != image features"
            else:
                assert n_feat == len(pk.image_rows), "image tokens != image features"
            N_pad = ix.round_up(plan.n_patches, 256)      # N_pad/4 (merger rows) stays a multiple of 64
            gather = np.zeros(N_pad, dtype=np.int32); gather[:plan.n_patches] = plan.patch_gather
            vcos = np.zeros((N_pad, c.v_head_dim // 2), np.float32); vcos[:plan.n_patches] = plan.cos
            vsin = np.zeros_like(vcos); vsin[:plan.n_patches] = plan.sin
            px = pixel_values if torch.is_tensor(pixel_values) else torch.from_numpy(np.asarray(pixel_values))
            sq = lambda cu: float((np.diff(np.asarray(cu, dtype=np.float64)) ** 2).sum())
            vis = dict(pairs_win=sq(plan.cu_window), pairs_img=sq(plan.cu_image), N=plan.n_patches, N_pad=N_pad, px=px.to(device=dev, dtype=F32, non_blocking=True), gather=t(gather, I32),
                       inverse=t(plan.merged_inverse, I32), cu_win=t(plan.cu_window, I32), cu_img=t(plan.cu_image, I32),
                       max_win=plan.max_window, max_img=plan.max_image, cos=t(vcos, F32), sin=t(vsin, F32),
                       img_src=None if img_src is None else t(img_src, I32), dup_idx=None if dup_idx is None else t(dup_idx, I32))
        seg = tuple(t(a, I32) for a in (pk.seg_b, pk.seg_e, pk.pre_b, pk.pre_e, pk.dep_e))
        return DeviceBatch(pk, t(pk.ids, I32), t(pk.embed_ids, I32), t(pk.cu_seqlens, I32), cos, sin, t(pk.image_rows, I32),
                           t(rows, I32), t(labels, I64), t(pk.out_index, I64), Tr_pad, vis, seg,
                           None if pk.logit_dup is None else t(pk.logit_dup, I32),
                           None if pk.logit_distinct is None else t(pk.logit_distinct, I32), pairs=_seg_pairs(pk))

    # ---------------------------------------------------------------- vision tower
    # ViT attention has head dim 80.  Whole-image blocks (1344 patches at the STVQA shape, 4096 at 896 px) can run on the D = 128 kernels
    # (LDS-DMA staged, MFMA 32x32x16) with heads zero-padded to 128: the pad contributes 0 to q.k and its V columns are 0, so the first 80
    # output dims are the same attention at 1.6x the flops.  Measured incl. the pad / slice copies (tools/vit_attn_ab.py, TF/s on the
    # D = 80 flops): backward 131-144 -> 222-372 at 1024..4096 patches; forward 321 -> 320 at 1344, 353 -> 468 at 4096, slower at 1024.
    # Hence two thresholds; forward and backward may use different kernels (lse and O agree within the kernels' own bf16 noise).
    VIT_PAD_MIN_SEQ = 2048          # forward
    VIT_PAD_MIN_SEQ_BWD = 512       # backward

    def _vit_pad(self, x3: torch.Tensor) -> torch.Tensor:
        """(Np, k, heads, 80) view -> contiguous (Np, k*heads*128) with zero pad"""
        Np, k, heads, hd = x3.shape
        out = torch.zeros(Np, k, heads, 128, dtype=BF16, device=x3.device)
        out[..., :hd] = x3
        return out.view(Np, k * heads * 128)

    def _vit_attn_fwd(self, qkv, cu, mx, a, pairs, tokens=None):
        c = self.cfg
        heads, hd, vh = c.v_heads, c.v_head_dim, c.v_hidden
        if hd < 128 and mx >= self.VIT_PAD_MIN_SEQ:
            Np, W = qkv.shape[0], heads * 128
            qp = self._vit_pad(qkv.view(Np, 3, heads, hd))
            op, lse = ops.attn_fwd(qp[:, :W], qp[:, W:2 * W], qp[:, 2 * W:], cu, mx, heads, heads, 128, self.v_scale, False,
                                   pairs=None if pairs is None else pairs * hd / 128.0)
            a.view(Np, heads, hd).copy_(op.view(Np, heads, 128)[..., :hd])
            return lse
        _, lse = ops.attn_fwd(qkv[:, :vh], qkv[:, vh:2 * vh], qkv[:, 2 * vh:], cu, mx, heads, heads, hd, self.v_scale, False, out=a, pairs=pairs, tokens=tokens)
        return lse

    def _vit_attn_bwd(self, qkv, a, da, lse, cu, mx, dqkv, pairs, tokens=None):
        c = self.cfg
        heads, hd, vh = c.v_heads, c.v_head_dim, c.v_hidden
        if hd < 128 and mx >= self.VIT_PAD_MIN_SEQ_BWD:
            Np, W = qkv.shape[0], heads * 128
            qp = self._vit_pad(qkv.view(Np, 3, heads, hd))
            ap, dap = self._vit_pad(a.view(Np, 1, heads, hd)), self._vit_pad(da.view(Np, 1, heads, hd))
            dp = torch.zeros(Np, 3 * W, dtype=BF16, device=qkv.device)
            ops.attn_bwd(qp[:, :W], qp[:, W:2 * W], qp[:, 2 * W:], ap, dap, lse, cu, mx, heads, heads, 128, self.v_scale, False,
                         dp[:, :W], dp[:, W:2 * W], dp[:, 2 * W:], pairs=None if pairs is None else pairs * hd / 128.0)
            dqkv.view(Np, 3, heads, hd).copy_(dp.view(Np, 3, heads, 128)[..., :hd])
            return
        ops.attn_bwd(qkv[:, :vh], qkv[:, vh:2 * vh], qkv[:, 2 * vh:], a, da, lse, cu, mx, heads, heads, hd, self.v_scale, False,
                     dqkv[:, :vh], dqkv[:, vh:2 * vh], dqkv[:, 2 * vh:], pairs=pairs, tokens=tokens)

    def _vit_forward(self, b: DeviceBatch, save: Optional[list]):
        c, w, v = self.cfg, self.p.w, b.vis
        N, Np = v["N"], v["N_pad"]
        heads, hd, vh = c.v_heads, c.v_head_dim, c.v_hidden
        px = ops.cast_pad(v["px"], c.patch_k_pad)                                    # (N, Kpad) bf16, pixel order
        pxw = torch.zeros(Np, c.patch_k_pad, dtype=BF16, device=px.device)
        ops.rows_gather(px, v["gather"][:N], out=pxw[:N])                            # window order (row gather commutes with the GEMM)
        x = ops.gemm_nt(pxw, w["v.patch_embed"])                                     # HF patch_embed :99-122
        if save is not None:
            save.append(("pe", pxw))
        for i in range(c.v_depth):
            p = f"v.{i}."
            h1, r1 = ops.rmsnorm_fwd(x, w[p + "norm1"], 1e-6)
            qkv = ops.gemm_nt(h1, w[p + "qkv_w"], bias=w[p + "qkv_b"])
            ops.rope_apply_(qkv, v["cos"], v["sin"], 2 * heads, hd)
            full = i in c.v_fullatt
            cu, mx = (v["cu_img"], v["max_img"]) if full else (v["cu_win"], v["max_win"])
            a = torch.zeros(Np, vh, dtype=BF16, device=x.device)
            lse = self._vit_attn_fwd(qkv, cu, mx, a, v["pairs_img"] if full else v["pairs_win"], tokens=N)
            x1 = ops.gemm_nt(a, w[p + "proj_w"], bias=w[p + "proj_b"], residual=x)
            h2, r2 = ops.rmsnorm_fwd(x1, w[p + "norm2"], 1e-6)
            gu = ops.gemm_nt(h2, w[p + "gu_w"], bias=w[p + "gu_b"])
            m = ops.swiglu_fwd(gu)
            x2 = ops.gemm_nt(m, w[p + "down_w"], bias=w[p + "down_b"], residual=x1)
            if save is not None:
                save.append((x, r1, h1, qkv, a, lse, x1, r2, h2, gu, m))
            x = x2
        hq, rq = ops.rmsnorm_fwd(x, w["v.merger.ln_q"], 1e-6)
        hm = hq.view(Np // 4, 4 * vh)
        f1 = ops.gemm_nt(hm, w["v.merger.fc1_w"], bias=w["v.merger.fc1_b"])
        ge = ops.gelu_fwd(f1)
        f2 = ops.gemm_nt(ge, w["v.merger.fc2_w"], bias=w["v.merger.fc2_b"])          # (Np/4, H) window order
        img = ops.rows_gather(f2, v["inverse"])                                       # back to image order (HF :468-470)
        if save is not None:
            save.append((x, rq, hq, f1, ge))
        return img

    def _vit_backward(self, b: DeviceBatch, saved: list, d_img: torch.Tensor):
        c, w, g, v = self.cfg, self.p.w, self.p.g, b.vis
        N, Np = v["N"], v["N_pad"]
        heads, hd, vh = c.v_heads, c.v_head_dim, c.v_hidden
        x, rq, hq, f1, ge = saved.pop()
        df2 = torch.zeros(Np // 4, c.hidden_size, dtype=BF16, device=d_img.device)
        ops.rows_scatter_(df2, v["inverse"], d_img)
        self._dw(g["v.merger.fc2_w"], df2, ge, g["v.merger.fc2_b"])
        dge = ops.gemm_nn(df2, w["v.merger.fc2_w"])
        df1 = ops.gelu_bwd(f1, dge)
        self._dw(g["v.merger.fc1_w"], df1, hq.view(Np // 4, 4 * vh), g["v.merger.fc1_b"])
        dhq = ops.gemm_nn(df1, w["v.merger.fc1_w"]).view(Np, vh)
        dx = ops.rmsnorm_bwd(x, w["v.merger.ln_q"], rq, dhq, dw_accum=g["v.merger.ln_q"])
        for i in reversed(range(c.v_depth)):
            p = f"v.{i}."
            x0, r1, h1, qkv, a, lse, x1, r2, h2, gu, m = saved.pop()
            self._dw(g[p + "down_w"], dx, m, g[p + "down_b"])
            dm = ops.gemm_nn(dx, w[p + "down_w"])
            dgu = ops.swiglu_bwd(gu, dm)
            self._dw(g[p + "gu_w"], dgu, h2, g[p + "gu_b"])
            dh2 = ops.gemm_nn(dgu, w[p + "gu_w"])
            dx1 = ops.rmsnorm_bwd(x1, w[p + "norm2"], r2, dh2, dres=dx, dw_accum=g[p + "norm2"])
            self._dw(g[p + "proj_w"], dx1, a, g[p + "proj_b"])
            da = ops.gemm_nn(dx1, w[p + "proj_w"])
            full = i in c.v_fullatt
            cu, mx = (v["cu_img"], v["max_img"]) if full else (v["cu_win"], v["max_win"])
            dqkv = torch.zeros_like(qkv)
            self._vit_attn_bwd(qkv, a, da, lse, cu, mx, dqkv, v["pairs_img"] if full else v["pairs_win"], tokens=N)
            ops.rope_apply_(dqkv, v["cos"], v["sin"], 2 * heads, hd, inverse=True)
            self._dw(g[p + "qkv_w"], dqkv, h1, g[p + "qkv_b"])
            dh1 = ops.gemm_nn(dqkv, w[p + "qkv_w"])
            dx = ops.rmsnorm_bwd(x0, w[p + "norm1"], r1, dh1, dres=dx1, dw_accum=g[p + "norm1"])
        (_, pxw) = saved.pop()
        self._dw(g["v.patch_embed"], dx, pxw, None)

    # ---------------------------------------------------------------- helpers
    def _dw(self, gw: torch.Tensor, dy: torch.Tensor, x: torch.Tensor, gb: Optional[torch.Tensor], fp8: bool = False, dyt=None):
        """gw (N,K) fp32 += dy(M,N)^T x(M,K); gb (N,) += column sums of dy.  M is a multiple of 64 (padded rows are zero).
        fp8 (enable_fp8(wgrad=True), LM projections): both operands quantised token-minor on the fly (st_mxfp8_quantize_t: MX blocks of 32
        consecutive packed tokens), the product on the 4-wave fp8 tile accumulating into the fp32 gradient."""
        if fp8 and dy.shape[0] > 256 and dy.shape[0] % 128 == 0:
            dq, ds = dyt if dyt is not None else ops.mxfp8_quantize_t(dy)
            xq, xs = ops.mxfp8_quantize_t(x)
            ops.gemm_mxfp8_nt_f32(dq, ds, xq, xs, gw, accumulate=True)
        elif ops.layout_gemm_ok(dy.shape[1], x.shape[1], dy.shape[0]):
            # both operands as they lie in memory (token index = contraction = row): transpose reads inside the tile, no transposed
            # copies of dY / X in HBM (+13 % on the layer's dW GEMMs vs transposes + NT, tools/gemm_layout_ab.py)
            ops.gemm_tn(dy, x, gw, accumulate=True)
        else:
            ops.gemm_nt(ops.transpose(dy), ops.transpose(x), out_f32=gw, accumulate=True)
        if gb is not None:
            ops.colsum(dy, out_f32=gb, accumulate=True)

    def _embed(self, b: DeviceBatch, save_vit: Optional[list]):
        x = ops.embed_gather(self.p.w["embed"], b.ids)
        if b.vis is not None:
            img = self._vit_forward(b, save_vit)
            if b.vis["img_src"] is not None:                                          # shared image: one feature row feeds several samples
                img = ops.rows_gather(img, b.vis["img_src"])
            ops.rows_scatter_(x, b.image_rows, img)                                   # masked_scatter, HF :1209-1215
        if b.pk.T_pad > b.pk.T:
            x[b.pk.T:].zero_()
        return x

    # ---------------------------------------------------------------- language model
    def _lm_layer_fwd(self, i: int, x0: torch.Tensor, b: DeviceBatch, save: Optional[list], kv_out=None, prefix_kv=None):
        c, w = self.cfg, self.p.w
        p = f"l.{i}."
        D, nq, nkv = c.head_dim, c.num_heads, c.num_kv_heads
        # fp8 mode: the producers of a projection's input emit its MX-fp8 operand in the same pass (RMSNorm, SwiGLU — bit-identical to the
        # bf16 op + st_mxfp8_quantize); the bf16 tensor itself is written only when the backward keeps it for the weight gradient
        fp8 = self.fp8 and x0.shape[0] > 256
        keep = save is not None and not self.recompute_light
        if fp8:
            h1, r1, h1q = ops.rmsnorm_mxfp8(x0, w[p + "in_norm"], c.rms_eps, want_y=keep, want_rstd=save is not None)
            qkv = self._linear(None, p + "qkv_w", bias=w[p + "qkv_b"], xq=h1q)
        else:
            h1, r1 = ops.rmsnorm_fwd(x0, w[p + "in_norm"], c.rms_eps)
            qkv = self._linear(h1, p + "qkv_w", bias=w[p + "qkv_b"])
        spx = None
        if self.sp > 1:
            # this rank's row slice through the projections; ALL rows of n_heads / sp heads through the attention kernel
            assert kv_out is None and prefix_kv is None and not fp8, "sequence parallelism covers the plain log-prob / update passes"
            lo, Tl = self._sp_rows(b.pk.T_pad)
            ops.rope_apply_(qkv, b.cos[lo:lo + Tl], b.sin[lo:lo + Tl], nq + nkv, D)
            qs = self._sp_gather_seq(qkv[:, :nq * D], nq)
            ks = self._sp_gather_seq(qkv[:, nq * D:(nq + nkv) * D], nkv)
            vs = self._sp_gather_seq(qkv[:, (nq + nkv) * D:], nkv)
            a_s = torch.zeros(b.pk.T_pad, (nq // self.sp) * D, dtype=BF16, device=x0.device)
            _, lse = ops.attn_fwd_seg(qs, ks, vs, b.seg[0], b.seg[1], b.seg[2], b.seg[3], b.pk.max_seg, nq // self.sp, nkv // self.sp, D,
                                      self.scale, out=a_s, pairs=b.pairs / self.sp)
            a = self._sp_scatter_seq(a_s, nq)
            spx = (qs, ks, vs, a_s)
        else:
            ops.rope_apply_(qkv, b.cos, b.sin, nq + nkv, D)
            q, k, v = qkv[:, :nq * D], qkv[:, nq * D:(nq + nkv) * D], qkv[:, (nq + nkv) * D:]
            if kv_out is not None:
                kv_out(i, k, v)
            a = torch.zeros(x0.shape[0], nq * D, dtype=BF16, device=x0.device)
            kpre, vpre = prefix_kv if prefix_kv is not None else (None, None)     # prompt K/V cached by the rollout prefill
            _, lse = ops.attn_fwd_seg(q, k, v, b.seg[0], b.seg[1], b.seg[2], b.seg[3], b.pk.max_seg, nq, nkv, D, self.scale, out=a,
                                      k_pre=kpre, v_pre=vpre, pairs=b.pairs)
        x1 = self._linear(a, p + "o_w", residual=x0)
        if fp8:
            h2, r2, h2q = ops.rmsnorm_mxfp8(x1, w[p + "post_norm"], c.rms_eps, want_y=keep, want_rstd=save is not None)
            wq, ws = self.p.wq[p + "gu_w"]
            if save is None:                                   # no-grad passes: SwiGLU + quantiser in the fp8 tile's epilogue — neither
                gu, m = None, None                             # gate|up nor the bf16 activation is ever stored
                mq = ops.gemm_mxfp8_swiglu_q(h2q[0], h2q[1], wq, ws)
            else:
                gu = ops.gemm_mxfp8_nt(h2q[0], h2q[1], wq, ws)
                m, mq = ops.swiglu_mxfp8(gu, want_out=keep)
            x2 = self._linear(None, p + "down_w", residual=x1, xq=mq)
        else:
            h2, r2 = ops.rmsnorm_fwd(x1, w[p + "post_norm"], c.rms_eps)
            if save is not None and self.unfused_swiglu_with_grad and h2.shape[0] > 256:
                # with gradients the backward needs gate|up, and a GEMM epilogue that stores m AND gate|up (192 KiB per tile instead of
                # 128) costs more than the stand-alone SwiGLU pass over the kept gate|up (tools/swiglu_ab.py, T = 21504: 4.45-4.54 ms
                # fused vs 3.90-4.05 + 0.42); bit-identical either way (tests/test_gpu_kernels.py)
                gu = ops.gemm_nt(h2, w[p + "gu_w"])
                m = ops.swiglu_fwd(gu)
            else:
                gu, m = ops.gemm_swiglu(h2, w[p + "gu_w"], want_gu=save is not None)  # SwiGLU in the epilogue (no-grad passes: m only)
        if not fp8:
            x2 = self._linear(m, p + "down_w", residual=x1)
        if save is not None:
            # h1, h2 (RMSNorm outputs) and m (SwiGLU output) are cheap row-wise functions of tensors that are kept anyway: with
            # recompute_light the backward recomputes them (bit-identical kernels) instead of holding 1/3 of the activation
            # memory (-33 %, +1 % time) — off by default, the 4-micro-batch pass fits with room to spare
            if self.recompute_light:
                save.append((x0, r1, None, qkv, a, lse, x1, r2, None, gu, None, spx))
            else:
                save.append((x0, r1, h1, qkv, a, lse, x1, r2, h2, gu, m, spx))
        return x2

    def _lm_layer_bwd(self, i: int, dx2: torch.Tensor, b: DeviceBatch, saved):
        c, w, g = self.cfg, self.p.w, self.p.g
        p = f"l.{i}."
        D, nq, nkv = c.head_dim, c.num_heads, c.num_kv_heads
        x0, r1, h1, qkv, a, lse, x1, r2, h2, gu, m, spx = saved
        q2, t2 = self._quantize_grad(dx2)
        dm = self._dgrad(dx2, p + "down_w", dyq=q2)
        if m is None:                                       # recompute the light activations (see _lm_layer_fwd): m in the pass that
            dgu, m = ops.swiglu_bwd(gu, dm, want_m=True)    # forms dgu from the same gate | up values (bit-identical to swiglu_fwd)
            h2, _ = ops.rmsnorm_fwd(x1, w[p + "post_norm"], c.rms_eps, want_rstd=False)
            h1, _ = ops.rmsnorm_fwd(x0, w[p + "in_norm"], c.rms_eps, want_rstd=False)
        else:
            dgu = ops.swiglu_bwd(gu, dm)
        self._dw(g[p + "down_w"], dx2, m, None, fp8=self.fp8_wgrad, dyt=t2)
        qg, tg = self._quantize_grad(dgu)
        self._dw(g[p + "gu_w"], dgu, h2, None, fp8=self.fp8_wgrad, dyt=tg)
        dh2 = self._dgrad(dgu, p + "gu_w", dyq=qg)
        dx1 = ops.rmsnorm_bwd(x1, w[p + "post_norm"], r2, dh2, dres=dx2, dw_accum=g[p + "post_norm"])
        q1, t1 = self._quantize_grad(dx1)
        self._dw(g[p + "o_w"], dx1, a, None, fp8=self.fp8_wgrad, dyt=t1)
        da = self._dgrad(dx1, p + "o_w", dyq=q1)
        dqkv = torch.zeros_like(qkv)
        if spx is not None:                                 # the forward's two all-to-alls, transposed: dO out, dQ / dK / dV back
            qs, ks, vs, a_s = spx
            lo, Tl = self._sp_rows(b.pk.T_pad)
            da_s = self._sp_gather_seq(da, nq)
            dqs, dks, dvs = torch.zeros_like(qs), torch.zeros_like(ks), torch.zeros_like(vs)
            ops.attn_bwd_seg(qs, ks, vs, a_s, da_s, lse, b.seg[0], b.seg[1], b.seg[2], b.seg[3], b.seg[4], b.pk.T, b.pk.max_seg, nq // self.sp,
                             nkv // self.sp, D, self.scale, dqs, dks, dvs, pairs=b.pairs / self.sp)
            dqkv[:, :nq * D] = self._sp_scatter_seq(dqs, nq)
            dqkv[:, nq * D:(nq + nkv) * D] = self._sp_scatter_seq(dks, nkv)
            dqkv[:, (nq + nkv) * D:] = self._sp_scatter_seq(dvs, nkv)
            ops.rope_apply_(dqkv, b.cos[lo:lo + Tl], b.sin[lo:lo + Tl], nq + nkv, D, inverse=True)
        else:
            q, k, v = qkv[:, :nq * D], qkv[:, nq * D:(nq + nkv) * D], qkv[:, (nq + nkv) * D:]
            ops.attn_bwd_seg(q, k, v, a, da, lse, b.seg[0], b.seg[1], b.seg[2], b.seg[3], b.seg[4], b.pk.T, b.pk.max_seg, nq, nkv, D,
                             self.scale, dqkv[:, :nq * D], dqkv[:, nq * D:(nq + nkv) * D], dqkv[:, (nq + nkv) * D:], pairs=b.pairs)
            ops.rope_apply_(dqkv, b.cos, b.sin, nq + nkv, D, inverse=True)
        qq, tq = self._quantize_grad(dqkv)
        self._dw(g[p + "qkv_w"], dqkv, h1, g[p + "qkv_b"], fp8=self.fp8_wgrad, dyt=tq)
        dh1 = self._dgrad(dqkv, p + "qkv_w", dyq=qq)
        return ops.rmsnorm_bwd(x0, w[p + "in_norm"], r1, dh1, dres=dx1, dw_accum=g[p + "in_norm"])

    def _head_fwd(self, x: torch.Tensor, b: DeviceBatch, temperature: float):
        c, w = self.cfg, self.p.w
        xr = ops.rows_gather(x, b.logit_rows)                       # only rows that predict a response token
        hn, rn = ops.rmsnorm_fwd(xr, w["final_norm"], c.rms_eps)
        head = w["embed"] if c.tie_word_embeddings else w["lm_head"]
        logits = ops.gemm_nt(hn, head)
        logp, lse = ops.logprob_fwd(logits, b.labels, temperature)
        return xr, hn, rn, logits, logp, lse

    # ---------------------------------------------------------------- log-probs on top of the rollout's prompt cache
    def stage_responses(self, input_ids, attention_mask, position_ids, response_length: int, prompt_of_row, prompt_offsets) -> DeviceBatch:
        c, dev = self.cfg, self.p.device
        pk = ix.pack_responses(_np(input_ids), _np(attention_mask), _np(position_ids), response_length, prompt_of_row, prompt_offsets)
        t = lambda a, dt: ops.h2d(a, dt, dev)
        cos, sin = ops.mrope_table(t(pk.pos, I32), self.inv_freq, c.head_dim, c.mrope_section)
        n_first, Tr = len(pk.first_prompt), len(pk.labels)
        Tr_pad = max(ix.round_up(Tr, 128), 128)
        labels = np.full(Tr_pad, -1, dtype=np.int64); labels[:Tr] = pk.labels
        seg = tuple(t(a, I32) for a in (pk.seg_b, pk.seg_e, pk.pre_b, pk.pre_e, pk.dep_e))
        b = DeviceBatch(pk, t(pk.ids, I32), t(pk.embed_ids, I32), t(pk.cu_seqlens, I32), cos, sin, t(pk.image_rows, I32),
                        t(pk.logit_rows, I32), t(labels, I64), t(pk.out_index, I64), Tr_pad, None, seg, pairs=_seg_pairs(pk))
        b.first_prompt = t(pk.first_prompt, I32)
        return b

    @torch.no_grad()
    def log_probs_cached(self, b: DeviceBatch, cache: dict, temperature: float = 1.0) -> torch.Tensor:
        """(B, R) log-probs of the response tokens computed on the RESPONSE tokens only: the prompt part comes from `cache`
        (kp/vp (L, Tp, n_kv*D) prompt K/V and last_h (n_prompts, H) left by Generator.generate with the SAME weights) — what
        the no-grad pass would recompute, bit for bit."""
        c, w = self.cfg, self.p.w
        x = ops.embed_gather(w["embed"], b.ids)
        if b.pk.T_pad > b.pk.T:
            x[b.pk.T:].zero_()
        for i in range(c.num_layers):
            x = self._lm_layer_fwd(i, x, b, None, prefix_kv=(cache["kp"][i], cache["vp"][i]))
        n_first, n_rest = b.first_prompt.numel(), b.logit_rows.numel()
        xr = torch.zeros(b.Tr_pad, c.hidden_size, dtype=BF16, device=x.device)
        if n_first:
            ops.rows_gather(cache["last_h"], b.first_prompt, out=xr[:n_first])
        if n_rest:
            ops.rows_gather(x, b.logit_rows, out=xr[n_first:n_first + n_rest])
        hn, _ = ops.rmsnorm_fwd(xr, w["final_norm"], c.rms_eps)
        head = w["embed"] if c.tie_word_embeddings else w["lm_head"]
        logits = ops.gemm_nt(hn, head)
        logp, _ = ops.logprob_fwd(logits, b.labels, temperature)
        out = torch.zeros(b.pk.B * b.pk.R, dtype=F32, device=x.device)
        out.index_copy_(0, b.out_index, logp[:len(b.out_index)])
        return out.view(b.pk.B, b.pk.R)


    # ---------------------------------------------------------------- public entry points
    @torch.no_grad()
    def log_probs(self, b: DeviceBatch, temperature: float = 1.0) -> torch.Tensor:
        """(B, R) fp32 log-probs of the response tokens — DataParallelPPOActor._forward_micro_batch
        (verl/workers/actor/dp_actor.py:64-153), no-grad use."""
        x = self._embed(b, None)
        if self.sp > 1:
            lo, Tl = self._sp_rows(b.pk.T_pad)
            x = x[lo:lo + Tl].contiguous()
        for i in range(self.cfg.num_layers):
            x = self._lm_layer_fwd(i, x, b, None)
        if self.sp > 1:
            x = self._sp_all_gather_rows(x)                 # head + log-prob on the full stream (replicated, see set_sequence_parallel)
        *_, logp, _ = self._head_fwd(x, b, temperature)
        out = torch.zeros(b.pk.B * b.pk.R, dtype=F32, device=x.device)
        out.index_copy_(0, b.out_index, logp[:len(b.out_index)])
        return out.view(b.pk.B, b.pk.R)

    # ---------------------------------------------------------------- critic (cfg.value_head): dp_critic.py
    def _value_head_fwd(self, x: torch.Tensor, b: DeviceBatch):
        """values at the rows that PRECEDE a response token (`values[:, -R-1:-1]`, dp_critic.py:113,125 — the rows the actor's lm_head runs on)."""
        c, w = self.cfg, self.p.w
        xr = ops.rows_gather(x, b.logit_rows)
        hn, rn = ops.rmsnorm_fwd(xr, w["final_norm"], c.rms_eps)
        v = ops.value_head_fwd(hn, w["score_w"], w["score_b"])
        return xr, hn, rn, v

    @torch.no_grad()
    def values(self, b: DeviceBatch) -> torch.Tensor:
        """(B, R) fp32 value predictions — DataParallelPPOCritic._forward_micro_batch (dp_critic.py:52-125), no-grad use."""
        assert self.cfg.value_head and self.sp == 1, "the critic passes are not sequence-parallel (worker.critic.ulysses_sequence_parallel_size = 1)"
        x = self._embed(b, None)
        for i in range(self.cfg.num_layers):
            x = self._lm_layer_fwd(i, x, b, None)
        *_, v = self._value_head_fwd(x, b)
        out = torch.zeros(b.pk.B * b.pk.R, dtype=F32, device=x.device)
        out.index_copy_(0, b.out_index, v[:len(b.out_index)])
        return out.view(b.pk.B, b.pk.R)

    @torch.no_grad()
    def value_forward_backward(self, b: DeviceBatch, loss_inputs: dict, *, cliprange_value: float, grad_accum: float, loss_rows: Optional[int] = None,
                               on_final=None, train_vision: bool = True):
        """One micro-batch of update_critic (dp_critic.py:184-205): forward, clipped value loss, backward into the flat fp32 gradient
        buffer (accumulating).  loss_inputs: values, returns (B, R) fp32 and action_mask (B, R) int64 = attention_mask[:, -R-1:-1].
        Returns (vpreds (B, R), metrics (4,) or (k, 4) device: vf_loss, vf_clipfrac, masked mean of vpreds, mask count)."""
        c, g = self.cfg, self.p.g
        assert c.value_head
        off = self.p.offsets
        layer_lo = lambda i: off[f"l.{i}.in_norm"] if i < c.num_layers else off["final_norm"]
        vit_saved: Optional[list] = [] if train_vision else None
        x = self._embed(b, vit_saved)
        saved = []
        for i in range(c.num_layers):
            x = self._lm_layer_fwd(i, x, b, saved)
        xr, hn, rn, v = self._value_head_fwd(x, b)
        Tr, n = len(b.out_index), b.pk.B * b.pk.R
        vp_full = torch.zeros(n, dtype=F32, device=x.device)
        vp_full.index_copy_(0, b.out_index, v[:Tr])
        flat = lambda k: loss_inputs[k].reshape(-1).contiguous()
        val_f, ret_f, msk_f = flat("values").float(), flat("returns").float(), flat("action_mask").to(I64)
        loss_rows = loss_rows or b.pk.B
        if loss_rows >= b.pk.B:
            gfull, metrics = ops.value_loss(vp_full, ret_f, val_f, msk_f, cliprange_value=cliprange_value, grad_accum=grad_accum)
        else:                                                   # several reference micro-batches in one pass: each keeps its own masked means
            gs, ms = [], []
            for r0 in range(0, b.pk.B, loss_rows):
                sl = slice(r0 * b.pk.R, min(b.pk.B, r0 + loss_rows) * b.pk.R)
                g_i, m_i = ops.value_loss(vp_full[sl], ret_f[sl], val_f[sl], msk_f[sl], cliprange_value=cliprange_value, grad_accum=grad_accum)
                gs.append(g_i); ms.append(m_i)
            gfull, metrics = torch.cat(gs), torch.stack(ms)
        grow = torch.zeros(b.Tr_pad, dtype=F32, device=x.device)
        grow[:Tr] = gfull.index_select(0, b.out_index)
        dhn = ops.value_head_bwd(hn, self.p.w["score_w"], grow, g["score_w"], g["score_b"])
        dxr = ops.rmsnorm_bwd(xr, self.p.w["final_norm"], rn, dhn, dw_accum=g["final_norm"])
        if on_final is not None:                               # final_norm + score head close the buffer (param_layout)
            on_final(off["final_norm"], self.p.numel)
        dx = torch.zeros_like(x)
        if b.logit_dup is not None:
            ops.rows_scatter_(dx, b.logit_distinct, ops.rows_gather_sum(dxr, b.logit_dup))
        else:
            ops.rows_scatter_(dx, b.logit_rows[:Tr], dxr[:Tr])
        for i in reversed(range(c.num_layers)):
            dx = self._lm_layer_bwd(i, dx, b, saved.pop())
            if on_final is not None:
                on_final(layer_lo(i), layer_lo(i + 1))
        ops.embed_grad_(g["embed"], b.embed_ids, dx)
        if b.vis is not None and train_vision:
            d_img = ops.rows_gather(dx, b.image_rows)
            if b.vis["dup_idx"] is not None:
                d_img = ops.rows_gather_sum(d_img, b.vis["dup_idx"])
            self._vit_backward(b, vit_saved, d_img)
        return vp_full.view(b.pk.B, b.pk.R), metrics

    @torch.no_grad()
    def forward_backward(self, b: DeviceBatch, loss_inputs: dict, temperature: float = 1.0, **loss_kw):
        """One micro-batch of update_policy (dp_actor.py:240-278): forward, GRPO loss, backward into the
        flat fp32 gradient buffer (accumulating).  loss_inputs: old_log_probs, advantages, [ref_log_probs],
        response_mask — all (B, R) device tensors.  Returns (log_probs (B,R), metrics (8,) device).
        on_final(lo, hi): called as soon as the element range [lo, hi) of the flat gradient buffer has received its last
        contribution of THIS pass (head + final norm, then each LM layer in backward order) — the data-parallel engine starts
        that slice's all-reduce there when the pass is the last one of an optimizer step (actor.GradReducer)."""
        c, g, w = self.cfg, self.p.g, self.p.w
        on_final = loss_kw.pop("on_final", None)
        train_vision = loss_kw.pop("train_vision", True)     # False: frozen vision tower — no ViT activations kept, no ViT backward
        off = self.p.offsets
        layer_lo = lambda i: off[f"l.{i}.in_norm"] if i < c.num_layers else off["final_norm"]
        vit_saved: Optional[list] = [] if train_vision else None
        x = self._embed(b, vit_saved)
        if self.sp > 1:
            sp_lo, sp_Tl = self._sp_rows(b.pk.T_pad)
            x = x[sp_lo:sp_lo + sp_Tl].contiguous()
        saved = []
        for i in range(c.num_layers):
            x = self._lm_layer_fwd(i, x, b, saved)
        if self.sp > 1:
            x = self._sp_all_gather_rows(x)
        xr, hn, rn, logits, logp, lse = self._head_fwd(x, b, temperature)
        Tr, n = len(b.out_index), b.pk.B * b.pk.R
        lp_full = torch.zeros(n, dtype=F32, device=x.device)
        lp_full.index_copy_(0, b.out_index, logp[:Tr])
        flat = lambda k: loss_inputs[k].reshape(-1).contiguous()
        ref = flat("ref_log_probs") if loss_inputs.get("ref_log_probs") is not None else None
        # loss_rows: the reference's micro-batch size when several of its micro-batches ride in one pass (PolicyEngine fuses
        # them so a whole rollout group shares its prompt): every block of loss_rows rows gets its own masked means, exactly as if
        # it had been a separate forward/backward — gradients add up linearly in the shared backward.
        loss_rows = loss_kw.pop("loss_rows", None) or b.pk.B
        old_f, adv_f, msk_f = flat("old_log_probs").float(), flat("advantages").float(), flat("response_mask").to(I64)
        ref_f = ref.float() if ref is not None else None
        if loss_rows >= b.pk.B:
            gfull, metrics = ops.grpo_loss(lp_full, old_f, ref_f, adv_f, msk_f, **loss_kw)
        else:
            gs, ms = [], []
            for r0 in range(0, b.pk.B, loss_rows):
                sl = slice(r0 * b.pk.R, min(b.pk.B, r0 + loss_rows) * b.pk.R)
                g_i, m_i = ops.grpo_loss(lp_full[sl], old_f[sl], None if ref_f is None else ref_f[sl], adv_f[sl], msk_f[sl], **loss_kw)
                gs.append(g_i); ms.append(m_i)
            gfull, metrics = torch.cat(gs), torch.stack(ms)
        grow = torch.zeros(b.Tr_pad, dtype=F32, device=x.device)
        grow[:Tr] = gfull.index_select(0, b.out_index)
        if self.sp > 1:
            # the head ran on the full stream on every sp rank; only the logit rows that sit on THIS rank's token rows enter its backward,
            # so every weight gradient below is a partial sum over the rank's tokens (the sp ranks' gradients add up to the full one)
            mine = (b.logit_rows[:Tr] >= sp_lo) & (b.logit_rows[:Tr] < sp_lo + sp_Tl)
            grow[:Tr] *= mine.to(grow.dtype)
        ops.logprob_bwd_(logits, b.labels, lse, grow, temperature)                       # logits buffer now holds dlogits
        head_name = "embed" if c.tie_word_embeddings else "lm_head"
        self._dw(g[head_name], logits, hn, None)
        dhn = ops.gemm_nn(logits, w[head_name])
        dxr = ops.rmsnorm_bwd(xr, self.p.w["final_norm"], rn, dhn, dw_accum=g["final_norm"])
        if on_final is not None and not c.tie_word_embeddings:      # final_norm + lm_head close the buffer (param_layout)
            on_final(off["final_norm"], self.p.numel)
        dx = torch.zeros_like(x)
        if b.logit_dup is not None:                         # the last prompt row of a group predicts every member's first token
            ops.rows_scatter_(dx, b.logit_distinct, ops.rows_gather_sum(dxr, b.logit_dup))
        else:
            ops.rows_scatter_(dx, b.logit_rows[:Tr], dxr[:Tr])
        if self.sp > 1:
            dx = dx[sp_lo:sp_lo + sp_Tl].contiguous()
        for i in reversed(range(c.num_layers)):
            dx = self._lm_layer_bwd(i, dx, b, saved.pop())
            if on_final is not None:
                on_final(layer_lo(i), layer_lo(i + 1))
        if self.sp > 1:                                     # back on the full stream, zero outside this rank's rows: embedding and ViT
            dx_full = torch.zeros(b.pk.T_pad, dx.shape[1], dtype=dx.dtype, device=dx.device)      # gradients are partial sums as well
            dx_full[sp_lo:sp_lo + sp_Tl] = dx
            dx = dx_full
        ops.embed_grad_(g["embed"], b.embed_ids, dx)
        if b.vis is not None and train_vision:
            d_img = ops.rows_gather(dx, b.image_rows)
            if b.vis["dup_idx"] is not None:                                          # gradient of the shared features = sum over their users
                d_img = ops.rows_gather_sum(d_img, b.vis["dup_idx"])
            self._vit_backward(b, vit_saved, d_img)
        return lp_full.view(b.pk.B, b.pk.R), metrics


# ---- one host -> device copy per image and step.  The phases of a step (rollout prefill, reference pass, update passes) stage the SAME
# per-sample pixel tensors (6.3 MB of fp32 patches per STVQA image, pageable host memory) — three copies of 400 MB per GPU and step in the
# bench.  The device copy is kept, keyed by the host tensor's storage address + shape + dtype + version (the host tensor is held, so the
# address stays valid), until the next rollout starts (Generator.generate drops the cache: new prompts) or the cache holds more than
# ST_PIXEL_CACHE_MB.  The cache is module-global: actor and critic workers of one process share it (same images, same step).
_PIXEL_CACHE: Dict[tuple, tuple] = {}
_PIXEL_CACHE_BYTES = [0]


def pixels_on_device(t, device) -> torch.Tensor:
    """Device copy of one image's patch tensor.  Cached only for host TORCH tensors (the trainer path passes them: dataset rows keep
    the tensors alive for the step), keyed by storage address, shape, dtype AND the tensor's in-place version counter — an in-place edit of
    the host tensor is a miss, not stale pixels.  Anything else (numpy arrays: a fresh wrapper per call could never hit) is copied uncached."""
    if not torch.is_tensor(t):
        return ops.h2d(t, F32, device)
    if t.is_cuda:
        return t
    key = (t.data_ptr(), tuple(t.shape), str(t.dtype), int(t._version), str(device))
    hit = _PIXEL_CACHE.get(key)
    if hit is None:
        if _PIXEL_CACHE_BYTES[0] > int(os.environ.get("ST_PIXEL_CACHE_MB", "4096")) << 20:
            drop_pixel_cache()
        hit = (t, ops.h2d(t, F32, device))
        _PIXEL_CACHE[key] = hit
        _PIXEL_CACHE_BYTES[0] += hit[1].numel() * 4
    return hit[1]


def drop_pixel_cache():
    _PIXEL_CACHE.clear()
    _PIXEL_CACHE_BYTES[0] = 0


def _seg_pairs(pk) -> float:
    """(query, key) pairs of the shared-prefix causal attention over the packed segments: own rows causally + the whole prefix."""
    L = (pk.seg_e - pk.seg_b).astype(np.float64)
    return float((L * (L + 1) / 2 + L * (pk.pre_e - pk.pre_b)).sum())


def _np(x):
    if x is None:
        return None
    if torch.is_tensor(x):
        return x.detach().cpu().numpy()
    return np.asarray(x)
