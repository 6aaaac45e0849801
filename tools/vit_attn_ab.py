"""Whole-image ViT attention (head dim 80, 16 heads): the D = 80 kernels vs the D = 128 kernels on zero-padded heads (incl. the pad / slice copies).
    python tools/vit_attn_ab.py"""
import os, sys, types
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatialthinker_amd import ops
from spatialthinker_amd.model import Qwen25VL
heads, hd = 16, 80
W = heads * hd
def bench(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for L, imgs in ((1344, 64), (4096, 8), (1024, 32)):
    T = L * imgs
    qkv = (torch.randn(T, 3 * W, device="cuda") * 0.5).bfloat16()
    cu = torch.arange(0, T + 1, L, dtype=torch.int32, device="cuda")
    do = (torch.randn(T, W, device="cuda") * 0.5).bfloat16()
    me = types.SimpleNamespace(cfg=types.SimpleNamespace(v_heads=heads, v_head_dim=hd, v_hidden=W), v_scale=hd ** -0.5, VIT_PAD_MIN_SEQ=512, VIT_PAD_MIN_SEQ_BWD=512,
                               _vit_pad=lambda x: Qwen25VL._vit_pad(None, x))
    a = torch.zeros(T, W, dtype=torch.bfloat16, device="cuda"); dq = torch.zeros_like(qkv)
    lse = Qwen25VL._vit_attn_fwd(me, qkv, cu, L, a, None)
    t_pf = bench(lambda: Qwen25VL._vit_attn_fwd(me, qkv, cu, L, a, None))
    t_pb = bench(lambda: Qwen25VL._vit_attn_bwd(me, qkv, a, do, lse, cu, L, dq, None))
    a80, lse80 = ops.attn_fwd(qkv[:, :W], qkv[:, W:2 * W], qkv[:, 2 * W:], cu, L, heads, heads, hd, hd ** -0.5, False)
    t_f = bench(lambda: ops.attn_fwd(qkv[:, :W], qkv[:, W:2 * W], qkv[:, 2 * W:], cu, L, heads, heads, hd, hd ** -0.5, False, out=a))
    t_b = bench(lambda: ops.attn_bwd(qkv[:, :W], qkv[:, W:2 * W], qkv[:, 2 * W:], a80, do, lse80, cu, L, heads, heads, hd, hd ** -0.5, False, dq[:, :W], dq[:, W:2 * W], dq[:, 2 * W:]))
    fl = 4.0 * hd * heads * L * L * imgs
    print(f"L={L} x {imgs} images: fwd D=80 {t_f:8.0f} us ({fl / t_f / 1e6:5.0f} TF/s)  padded-128 {t_pf:8.0f} us ({fl / t_pf / 1e6:5.0f} TF/s) | bwd D=80 {t_b:8.0f} us ({2.5 * fl / t_b / 1e6:5.0f})  padded-128 {t_pb:8.0f} us ({2.5 * fl / t_pb / 1e6:5.0f})", flush=True)
