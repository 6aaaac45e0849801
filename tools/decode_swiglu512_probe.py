"""The one-pass 512-row gate/up + SwiGLU tile (gemm_swiglu512.hip, plan 512) against the two-round 256x160 tile (plan 1): bit identity at
ragged shapes, then microseconds per launch over rotating weight copies (beyond the 256-MB MALL), hipGraph-replayed like the decode loop.
    python tools/decode_swiglu512_probe.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatialthinker_amd import ops
from spatialthinker_amd.lib import lib

L = lib()
EXTRA = int(os.environ.get("PROBE_VARIANT", "0"))      # a third variant id to check and time (experiments)


def run(variant, a, w, out):
    M, K = a.shape
    I = w.shape[0] // 2
    L.st_gemm_swiglu_decode_variant(variant, a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), out.data_ptr(), out.stride(0), M, I, K,
                                    torch.cuda.current_stream().cuda_stream)


torch.manual_seed(0)
for (M, I, K) in [(257, 1000, 128), (300, 80, 64), (512, 81, 192), (384, 18944, 3584), (512, 18944, 3584), (448, 11008, 2048), (1, 160, 64), (64, 240, 128)]:
    a = (torch.randn(M, K, device="cuda") * 0.5).bfloat16()
    w = (torch.randn(2 * I, K, device="cuda") * 0.05).bfloat16()
    o1 = torch.full((M, I), 7.0, device="cuda", dtype=torch.bfloat16); o5 = torch.full((M, I), -7.0, device="cuda", dtype=torch.bfloat16)
    run(1, a, w, o1); run(512, a, w, o5)
    if EXTRA:
        ox = torch.full((M, I), -3.0, device="cuda", dtype=torch.bfloat16); run(EXTRA, a, w, ox)
        print(f"   variant {EXTRA} identical to plan 1: {bool(torch.equal(o1, ox))}")
    torch.cuda.synchronize()
    want = a.float() @ w.float().t()
    g_, u_ = want[:, :I].bfloat16().float(), want[:, I:].bfloat16().float()
    ref = ((g_ * torch.sigmoid(g_)).bfloat16().float() * u_)
    e5 = float((o5.float() - ref).abs().max() / ref.abs().max())
    print(f"M={M} I={I} K={K}: identical to plan 1: {bool(torch.equal(o1, o5))}  (differing elements {int((o1 != o5).sum())}), rel err vs fp32 {e5:.4f}", flush=True)

I, K = 18944, 3584
ws = [(torch.randn(2 * I, K, device="cuda") * 0.05).bfloat16() for _ in range(8)]
for M in (257, 320, 384, 448, 512):
    a = (torch.randn(M, K, device="cuda") * 0.5).bfloat16()
    out = torch.empty(M, I, device="cuda", dtype=torch.bfloat16)
    res = {}
    for variant in (1, 512, 1, 512) + ((EXTRA, EXTRA) if EXTRA else ()):
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
    fl = 2.0 * M * 2 * I * K
    print(f"M={M}: plan 1 (256x160, two rounds) {min(res[1]):.1f} us | plan 512 (one pass) {min(res[512]):.1f} us = {fl / min(res[512]) * 1e-6:.0f} TF/s"
          + (f" | variant {EXTRA}: {min(res[EXTRA]):.1f} us" if EXTRA else ""), flush=True)
