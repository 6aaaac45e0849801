"""A/B of the direct (variant 6) and LDS-staged (variant 23) epilogues on the 7B layer's forward / dX / dW shapes, interleaved
rounds in one process (median of 7): python tools/gemm_epi_ab.py [T]"""
import os, sys, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("ST_LIB"):
    import spatialthinker_amd.lib as _lib
    _lib.LIB_PATH = os.path.abspath(os.environ["ST_LIB"])
from spatialthinker_amd import ops
T = int(sys.argv[1]) if len(sys.argv) > 1 else 10496
VA, VB = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (6, 23)
H, QKV, I2, I = 3584, 4608, 37888, 18944
ops._gemm_workspace(torch.device("cuda"))
def timeit(fn, iters=6):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
shapes = [("qkv+bias", T, QKV, H, "bias"), ("o+res", T, H, H, "res"), ("down+res", T, H, I, "res"), ("dx_qkv", T, H, QKV, ""), ("dx_o", T, H, H, ""),
          ("dx_gu", T, H, I2, ""), ("dx_down", T, I, H, ""), ("dw_qkv", QKV, H, T, "acc"), ("dw_o", H, H, T, "acc"), ("dw_gu", I2, H, T, "acc"),
          ("dw_down", H, I, T, "acc"), ("lm_head", 2048, 152064, H, "")]
tot = {VA: 0.0, VB: 0.0}
for name, M, N, K, kind in shapes:
    a = torch.randn(M, K, device="cuda").bfloat16(); b = torch.randn(N, K, device="cuda").bfloat16()
    bias = torch.randn(N, device="cuda").bfloat16(); res = torch.randn(M, N, device="cuda").bfloat16()
    c = torch.empty(M, N, device="cuda", dtype=torch.bfloat16); f = torch.zeros(M, N, device="cuda") if kind == "acc" else None
    def run(v):
        if kind == "acc": ops.gemm_nt_variant(v, a, b, out_f32=f, accumulate=True)
        elif kind == "bias": ops.gemm_nt_variant(v, a, b, out=c, bias=bias)
        elif kind == "res": ops.gemm_nt_variant(v, a, b, out=c, residual=res)
        else: ops.gemm_nt_variant(v, a, b, out=c)
    for v in (VA, VB): run(v)
    ts = {VA: [], VB: []}
    for _ in range(7):
        for v in (VA, VB): ts[v].append(timeit(lambda: run(v)))
    m6, m23 = statistics.median(ts[VA]), statistics.median(ts[VB])
    tot[VA] += m6; tot[VB] += m23
    fl = 2.0 * M * N * K / 1e6
    print(f"{name:10s} {M:6d}x{N:6d}x{K:6d}: v{VA} {m6:8.1f} us ({fl / m6:5.0f} TF)   v{VB} {m23:8.1f} us ({fl / m23:5.0f} TF)   {100 * (m6 / m23 - 1):+5.1f} %", flush=True)
    del a, b, bias, res, c, f
print(f"sum: variant {VA} {tot[VA] / 1e3:.2f} ms, variant {VB} {tot[VB] / 1e3:.2f} ms ({100 * (tot[VA] / tot[VB] - 1):+.1f} %)")
