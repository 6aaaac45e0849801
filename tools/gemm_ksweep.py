"""Separate the K-loop rate from the per-tile overhead: M x N fixed at whole rounds of 256x256 tiles (7 rounds of 256 CUs), K swept.
time(K) = rounds * (overhead + K/64 * t_ktile): slope -> cycles per K-tile, intercept -> prologue + epilogue per tile.
    python tools/gemm_ksweep.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatialthinker_amd import ops

M, N = 28672, 4096                       # 112 x 16 = 1792 tiles = 7.0 rounds
ops._gemm_workspace(torch.device("cuda"))
res = {}
for K in (3584, 14336):
    a = (torch.randn(M, K, device="cuda") * 0.5).bfloat16(); w = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    cands = [("v23", lambda: ops.gemm_nt_variant(23, a, w, out=out)), ("v40", lambda: ops.gemm_nt_variant(40, a, w, out=out)),
             ("lib", lambda: torch.matmul(a, w.t(), out=out))]
    if len(sys.argv) > 1 and sys.argv[1] == "debug":
        cands = [("v40", lambda: ops.gemm_nt_variant(40, a, w, out=out)), ("lib", lambda: torch.matmul(a, w.t(), out=out))] + [(f"v4{d}", (lambda d=d: ops.gemm_nt_variant(40 + d, a, w, out=out))) for d in (1, 2, 3)]
    for tag, fn in cands:
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = max(4, int(40 * 3584 / K))
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        res[(tag, K)] = ms
        print(f"K={K:6d} {tag}: {ms*1e3:8.1f} us  {2.0*M*N*K/ms/1e9:6.0f} TF/s  per tile-round {ms*1e3/7:7.1f} us", flush=True)
for tag in sorted({t for t, _ in res}):
    t1, t2 = res[(tag, 3584)], res[(tag, 14336)]
    per_kt = (t2 - t1) / 7 / ((14336 - 3584) / 64) * 1e3
    over = t1 * 1e3 / 7 - per_kt * 56
    print(f"{tag}: {per_kt:.3f} us per K-tile, {over:.1f} us overhead per tile")
