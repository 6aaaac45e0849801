"""Does a decode GEMM stream faster when its weights sit in the 256-MB infinity cache?  Per shape: microseconds per launch (hipGraph
replay) with 8 rotating weight copies (HBM) and with ONE copy replayed (cache-warm when it fits), at M rows.
    python3 tools/mall_warm_probe.py [M]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatialthinker_amd import ops

M = int(sys.argv[1]) if len(sys.argv) > 1 else 32


def timed(fn_list, reps=6):
    for f in fn_list: f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for f in fn_list: f()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * len(fn_list))


for name, N, K in (("qkv", 4608, 3584), ("o", 3584, 3584), ("down", 3584, 18944), ("gate/up", 37888, 3584)):
    a = (torch.randn(M, K, device="cuda") * 0.5).bfloat16()
    ws = [(torch.randn(N, K, device="cuda") * 0.05).bfloat16() for _ in range(8)]
    if name == "gate/up":
        out = torch.empty(M, N // 2, device="cuda", dtype=torch.bfloat16)
        mk = lambda w: (lambda: ops.gemm_swiglu_decode(a, w, out=out))
    else:
        mk = lambda w: (lambda: ops.gemm_nt_decode_slabs(a, w))
    cold = timed([mk(w) for w in ws])
    warm = timed([mk(ws[0]) for _ in range(8)])
    mb = N * K * 2 / 1e6
    print(f"M={M} {name:8s} {mb:6.1f} MB of weights: rotating copies {cold:6.1f} us ({mb / cold:5.2f} TB/s)   one copy {warm:6.1f} us ({mb / warm:5.2f} TB/s)", flush=True)
