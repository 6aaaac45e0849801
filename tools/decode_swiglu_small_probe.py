"""Decode gate/up + SwiGLU at <= 128 rows: the one-tile-per-CU variants (8 = 64 x 160, 9 = 128 x 160; 237 workgroups) against the 296-tile
plans (7 = 64 x 128, 6 = 128 x 128): bit identity vs the 256 x 160 tile (plan 1), then microseconds per launch over rotating weight copies.
    python tools/decode_swiglu_small_probe.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatialthinker_amd.lib import lib
L = lib()


def run(variant, a, w, out):
    M, K = a.shape
    I = w.shape[0] // 2
    L.st_gemm_swiglu_decode_variant(variant, a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), out.data_ptr(), out.stride(0), M, I, K,
                                    torch.cuda.current_stream().cuda_stream)


torch.manual_seed(0)
for (M, I, K) in [(17, 1000, 128), (64, 80, 64), (33, 81, 192), (64, 18944, 3584), (128, 18944, 3584), (100, 11008, 2048), (1, 160, 64)]:
    a = (torch.randn(M, K, device="cuda") * 0.5).bfloat16()
    w = (torch.randn(2 * I, K, device="cuda") * 0.05).bfloat16()
    o1 = torch.full((M, I), 7.0, device="cuda", dtype=torch.bfloat16)
    run(1, a, w, o1)
    for v in ((8, 9) if M <= 64 else (9,)):
        o = torch.full((M, I), -7.0, device="cuda", dtype=torch.bfloat16)
        run(v, a, w, o)
        torch.cuda.synchronize()
        print(f"M={M} I={I} K={K}: variant {v} identical to plan 1: {bool(torch.equal(o1, o))} (differing {int((o1 != o).sum())})", flush=True)

I, K = 18944, 3584
ws = [(torch.randn(2 * I, K, device="cuda") * 0.05).bfloat16() for _ in range(8)]
for M in (16, 32, 64, 96, 128):
    a = (torch.randn(M, K, device="cuda") * 0.5).bfloat16()
    out = torch.empty(M, I, device="cuda", dtype=torch.bfloat16)
    res = {}
    cands = (7, 8, 9) if M <= 64 else (6, 9)
    for variant in cands + cands:
        for w in ws:
            run(variant, a, w, out)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for w in ws:
                run(variant, a, w, out)
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            g.replay()
        e1.record(); torch.cuda.synchronize()
        res.setdefault(variant, []).append(e0.elapsed_time(e1) * 1e3 / 40)
    print(f"M={M}: " + " | ".join(f"variant {v}: {min(t):.1f} us ({2.0 * 2 * I * K / min(t) * 1e-6:.2f} TB/s of weights)" for v, t in res.items()), flush=True)
