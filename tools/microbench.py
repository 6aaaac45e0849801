#!/usr/bin/env python3
"""Per-kernel micro-benchmarks on one MI355X (not the headline bench — see bench.py).
Prints achieved TF/s or GB/s next to the roofline (2.5 PF bf16 MFMA dense, 8 TB/s HBM spec)."""
import json
import sys
import time

import torch

import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatialthinker_amd import ops  # noqa: E402


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    out = {}
    dev = "cuda"
    shapes = {"qkv7b": (8192, 4608, 3584), "o7b": (8192, 3584, 3584), "gateup7b": (8192, 37888, 3584),
              "down7b": (8192, 3584, 18944), "lmhead7b": (4096, 152064, 3584), "sq4096": (4096, 4096, 4096),
              "sq8192": (8192, 8192, 8192), "decode256": (256, 3584, 3584), "vit_qkv": (5376, 3840, 1280)}
    for name, (M, N, K) in shapes.items():
        a = torch.randn(M, K, device=dev).bfloat16()
        b = torch.randn(N, K, device=dev).bfloat16()
        c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        t_mine = timeit(lambda: ops.gemm_nt(a, b, out=c))
        t_blas = timeit(lambda: torch.matmul(a, b.t(), out=c))
        fl = 2.0 * M * N * K
        out[f"gemm_{name}"] = {"mine_TF": fl / t_mine / 1e12, "hipblaslt_TF": fl / t_blas / 1e12, "frac_peak": fl / t_mine / 2.5e15}
        print(name, out[f"gemm_{name}"], flush=True)
        del a, b, c
    # log-prob
    T, V = 4096, 152064
    z = torch.randn(T, V, device=dev).bfloat16()
    lab = torch.randint(0, V, (T,), device=dev)
    t = timeit(lambda: ops.logprob_fwd(z, lab, 1.0))
    out["logprob_fwd"] = {"GBs": 2.0 * T * V / t / 1e9, "frac_hbm": 2.0 * T * V / t / 8e12}
    logp, lse = ops.logprob_fwd(z, lab, 1.0)
    g = torch.randn(T, device=dev)
    t = timeit(lambda: ops.logprob_bwd_(z, lab, lse, g, 1.0))
    out["logprob_bwd"] = {"GBs": 4.0 * T * V / t / 1e9, "frac_hbm": 4.0 * T * V / t / 8e12}
    del z
    # rmsnorm
    T, H = 16384, 3584
    x = torch.randn(T, H, device=dev).bfloat16(); w = torch.ones(H, device=dev).bfloat16(); y = torch.empty_like(x)
    t = timeit(lambda: ops.rmsnorm_fwd(x, w, 1e-6, out=y))
    out["rmsnorm_fwd"] = {"GBs": 4.0 * T * H / t / 1e9, "frac_hbm": 4.0 * T * H / t / 8e12}
    # adamw
    n = 1 << 28
    p = torch.randn(n, device=dev).bfloat16(); gr = torch.randn(n, device=dev) * 1e-3
    m = torch.zeros_like(p); v = torch.zeros_like(p); c = torch.zeros_like(p)
    t = timeit(lambda: ops.adamw_kahan_step_(p, gr, m, v, c, t=3, lr=1e-6), iters=10)
    out["adamw"] = {"GBs": 20.0 * n / t / 1e9, "frac_hbm": 20.0 * n / t / 8e12}
    del p, gr, m, v, c
    # attention fwd (LM causal GQA 28/4, 4 sequences of 1612)
    for name, (lens, nq, nkv, D, causal) in {"lm7b": ([1612] * 8, 28, 4, 128, True), "vit_win": ([64] * 168, 16, 16, 80, False),
                                              "vit_full": ([1344] * 8, 16, 16, 80, False)}.items():
        Ttok = sum(lens)
        qkv = torch.randn(Ttok, (nq + 2 * nkv) * D, device=dev).bfloat16()
        cu = torch.tensor([0] + list(__import__("itertools").accumulate(lens)), dtype=torch.int32, device=dev)
        q, k, v = qkv[:, :nq * D], qkv[:, nq * D:(nq + nkv) * D], qkv[:, (nq + nkv) * D:]
        o = torch.empty(Ttok, nq * D, device=dev, dtype=torch.bfloat16)
        t = timeit(lambda: ops.attn_fwd(q, k, v, cu, max(lens), nq, nkv, D, D ** -0.5, causal, out=o))
        fl = sum(4.0 * D * L * L * nq for L in lens) * (0.5 if causal else 1.0)
        out[f"attn_fwd_{name}"] = {"TF": fl / t / 1e12, "ms": t * 1e3}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
