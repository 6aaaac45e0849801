"""A/B of the GEMM tile variants on the 7B shapes (interleaved rounds, random normal operands), with a correctness check."""
import sys
import torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatialthinker_amd import ops

def bench(fn, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3

variants = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0, 1, 2, 3, 4, 5]
shapes = {"qkv": (6528, 4608, 3584), "o": (6528, 3584, 3584), "gateup": (6528, 37888, 3584), "down": (6528, 3584, 18944),
          "dW_down": (3584, 18944, 6528), "lmhead": (2048, 152064, 3584), "sq4096": (4096, 4096, 4096), "sq8192": (8192, 8192, 8192),
          "exp26k": (26112, 4608, 3584)}
for name, (M, N, K) in shapes.items():
    a = torch.randn(M, K, device="cuda").bfloat16(); b = torch.randn(N, K, device="cuda").bfloat16()
    ref = torch.matmul(a, b.t())
    res = {}
    for rnd in range(2):
        for v in variants:
            c = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
            t = bench(lambda: ops.gemm_nt_variant(v, a, b, out=c))
            err = (c.float() - ref.float()).abs().max().item() / ref.float().abs().max().item()
            res.setdefault(v, []).append((2.0 * M * N * K / t / 1e12, err))
    tb = bench(lambda: torch.matmul(a, b.t()))
    line = " ".join(f"v{v}:{max(x[0] for x in r):7.0f}TF(err {max(x[1] for x in r):.1e})" for v, r in res.items())
    print(f"{name:8s} {M}x{N}x{K}: {line}  hipblaslt:{2.0 * M * N * K / tb / 1e12:7.0f}TF", flush=True)
    del a, b, ref
