"""Bit-compare two tile variants of st_gemm_nt_variant on a few shapes: python tools/gemm_variant_check.py vA vB"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatialthinker_amd import ops
VA, VB = int(sys.argv[1]), int(sys.argv[2])
ops._gemm_workspace(torch.device("cuda"))
torch.manual_seed(0)
for M, N, K, kind in [(1000, 512, 64, ""), (1000, 512, 128, ""), (777, 1000, 192, "bias"), (2048, 3584, 3584, "res"), (10496, 4608, 3584, "bias"), (3584, 4608, 2048, "acc"), (300, 300, 256, "")]:
    a = torch.randn(M, K, device="cuda").bfloat16(); b = torch.randn(N, K, device="cuda").bfloat16()
    bias = torch.randn(N, device="cuda").bfloat16(); res = torch.randn(M, N, device="cuda").bfloat16()
    outs = []
    for v in (VA, VB):
        c = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16); f = torch.ones(M, N, device="cuda")
        if kind == "acc": ops.gemm_nt_variant(v, a, b, out_f32=f, accumulate=True); outs.append(f)
        elif kind == "bias": ops.gemm_nt_variant(v, a, b, out=c, bias=bias); outs.append(c)
        elif kind == "res": ops.gemm_nt_variant(v, a, b, out=c, residual=res); outs.append(c)
        else: ops.gemm_nt_variant(v, a, b, out=c); outs.append(c)
    torch.cuda.synchronize()
    ref = a.float() @ b.float().t()
    print(f"{M}x{N}x{K} {kind or 'plain':5s}: identical = {torch.equal(outs[0], outs[1])}, max |vB - fp32 ref| = {(outs[1].float() - ref - (bias.float() if kind == 'bias' else 0) - (res.float() if kind == 'res' else 0) - (1 if kind == 'acc' else 0)).abs().max().item():.3f}")
