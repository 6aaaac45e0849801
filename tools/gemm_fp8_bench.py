"""MX-fp8 GEMM throughput on the 7B layer's forward shapes vs the bf16 kernel: python tools/gemm_fp8_bench.py [T]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatialthinker_amd import ops
T = int(sys.argv[1]) if len(sys.argv) > 1 else 10496
H, QKV, I2, I = 3584, 4608, 37888, 18944
def timeit(fn, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for name, M, N, K in [("qkv", T, QKV, H), ("o", T, H, H), ("gateup", T, I2, H), ("down", T, H, I), ("8192^3", 8192, 8192, 8192), ("16k qkv", 16384, QKV, H), ("16k gateup", 16384, I2, H)]:
    a = torch.randn(M, K, device="cuda").bfloat16(); b = torch.randn(N, K, device="cuda").bfloat16()
    c = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    aq, sa = ops.mxfp8_quantize(a); bq, sb = ops.mxfp8_quantize(b)
    ops.gemm_mxfp8_select(8)
    t8w = timeit(lambda: ops.gemm_mxfp8_nt(aq, sa, bq, sb, out=c))
    ops.gemm_mxfp8_select(4)
    t8 = timeit(lambda: ops.gemm_mxfp8_nt(aq, sa, bq, sb, out=c))
    t16 = timeit(lambda: ops.gemm_nt(a, b, out=c))
    tq = timeit(lambda: ops.mxfp8_quantize(a))
    fl = 2.0 * M * N * K / 1e6
    print(f"{name:8s} {M:6d}x{N:6d}x{K:6d}: mx-fp8 4-wave {t8:8.1f} us = {fl / t8:6.0f} TF ({fl / t8 / 5e3 * 100:4.1f} % of 5 PF)   8-wave {t8w:8.1f} us = {fl / t8w:6.0f} TF   bf16 {t16:8.1f} us = {fl / t16:6.0f} TF   "
          f"quantise A {tq:6.1f} us ({M * K * 3 / tq / 1e6:5.2f} TB/s)", flush=True)
