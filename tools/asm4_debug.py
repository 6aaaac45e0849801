"""Debug probe for gemm variant 40: which k positions / rows / columns of a 256x256x64 tile come out wrong."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatialthinker_amd import ops

torch.manual_seed(0)
M = N = 256
for K in (64, 128):
    bad_k = []
    for k0 in range(K):
        a = torch.zeros(M, K, device="cuda", dtype=torch.bfloat16); a[:, k0] = 1.0
        b = (torch.arange(K, device="cuda", dtype=torch.float32) + 1.0)[None, :].repeat(N, 1).bfloat16()
        c = ops.gemm_nt_variant(40, a, b).float()
        want = float(k0 + 1)
        wrong = (c != want)
        if wrong.any():
            rows = wrong.any(1).nonzero().flatten().tolist()
            cols = wrong.any(0).nonzero().flatten().tolist()
            vals = torch.unique(c[wrong]).tolist()[:8]
            bad_k.append(k0)
            if len(bad_k) <= 6:
                print(f"K={K} k0={k0}: {int(wrong.sum())} wrong; rows {rows[:8]}..{len(rows)} cols {cols[:8]}..{len(cols)} values {vals}")
    print(f"K={K}: bad k0 = {bad_k}")
# row / column identity patterns: A[m, k] = m-th row marker
K = 64
a = torch.zeros(M, K, device="cuda", dtype=torch.bfloat16)
a[:, 0] = torch.arange(M, device="cuda").bfloat16()                # exact up to 256
b = torch.zeros(N, K, device="cuda", dtype=torch.bfloat16); b[:, 0] = 1.0
c = ops.gemm_nt_variant(40, a, b).float()
want = torch.arange(M, device="cuda").float()[:, None].repeat(1, N)
print("row marker mismatches:", int((c != want).sum()), c[:4, :4].tolist(), c[128:132, 128:132].tolist())
a = torch.zeros(M, K, device="cuda", dtype=torch.bfloat16); a[:, 0] = 1.0
b = torch.zeros(N, K, device="cuda", dtype=torch.bfloat16); b[:, 0] = torch.arange(N, device="cuda").bfloat16()
c = ops.gemm_nt_variant(40, a, b).float()
want = torch.arange(N, device="cuda").float()[None, :].repeat(M, 1)
print("col marker mismatches:", int((c != want).sum()), c[:2, :8].tolist())
# random, error map
a = torch.randn(M, 64, device="cuda").bfloat16(); b = torch.randn(N, 64, device="cuda").bfloat16()
c = ops.gemm_nt_variant(40, a, b).float(); w = a.float() @ b.float().t()
e = (c - w).abs()
print("random K=64: max err", float(e.max()), "rows with err>0.1:", (e.max(1).values > 0.1).nonzero().flatten().tolist()[:20], "cols:", (e.max(0).values > 0.1).nonzero().flatten().tolist()[:20])
for rep in range(3):
    c2 = ops.gemm_nt_variant(40, a, b).float()
    print("repeatable:", bool(torch.equal(c, c2)))
