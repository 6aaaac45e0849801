"""ViT window attention, per launch: attention_win.hip against the generic D = 80 kernels (forced with max_seqlen = 65).
    python3 tools/vit_window_probe.py [images ...]     (windows of a 1344-patch image: 21 x 64 tokens; 16 heads)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from spatialthinker_amd import ops

def t_us(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n

heads, D = 16, 80
W = heads * D
for imgs in [int(a) for a in sys.argv[1:]] or [4, 16, 64]:
    lens = [64] * (21 * imgs)
    T = sum(lens)
    cu = torch.tensor(np.concatenate([[0], np.cumsum(lens)]).astype(np.int32), device="cuda")
    x = (torch.randn(T, 3 * W, device="cuda") * 0.5).bfloat16()
    q, k, v = x[:, :W], x[:, W:2 * W], x[:, 2 * W:]
    do = torch.randn(T, W, device="cuda").bfloat16()
    dqkv = torch.zeros_like(x)
    sc = D ** -0.5
    out, lse = ops.attn_fwd(q, k, v, cu, 64, heads, heads, D, sc, False)
    row = [f"{imgs:3d} images ({T} tokens)"]
    for mx in (64, 65):
        f = t_us(lambda: ops.attn_fwd(q, k, v, cu, mx, heads, heads, D, sc, False, out=out))
        b = t_us(lambda: ops.attn_bwd(q, k, v, out, do, lse, cu, mx, heads, heads, D, sc, False, dqkv[:, :W], dqkv[:, W:2 * W], dqkv[:, 2 * W:]))
        row.append(f"{'window' if mx == 64 else 'generic'} kernels: fwd {f:7.1f} us  bwd {b:7.1f} us")
    fl = 4.0 * D * 64 * 64 * heads * len(lens)
    by = T * W * 2 * 4
    row.append(f"fwd floor: {by / 8e12 * 1e6:.1f} us of HBM, {fl / 2.5e15 * 1e6:.2f} us of MFMA")
    print("   ".join(row), flush=True)
