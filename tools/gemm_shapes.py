"""All twelve GEMM shapes of one 7B LM layer (forward, dX, dW) at the micro-batch token counts of the bench: the dispatched
st_gemm_nt vs the two tile kernels (variant 0 = 128x128, 4 = 256x256) vs hipBLASLt (torch.matmul)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatialthinker_amd import ops  # noqa: E402


def bench(fn, iters=8):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


H, QKV, I2, I = 3584, 4608, 37888, 18944
for T in [int(a) for a in sys.argv[1:]] or [6528, 8704]:
    shapes = [("qkv", T, QKV, H), ("o", T, H, H), ("gateup", T, I2, H), ("down", T, H, I),
              ("dx_qkv", T, H, QKV), ("dx_o", T, H, H), ("dx_gu", T, H, I2), ("dx_down", T, I, H),
              ("dw_qkv", QKV, H, T), ("dw_o", H, H, T), ("dw_gu", I2, H, T), ("dw_down", H, I, T)]
    tot = {"disp": 0.0, "v0": 0.0, "v4": 0.0, "blas": 0.0}
    for name, M, N, K in shapes:
        a = torch.randn(M, K, device="cuda").bfloat16(); b = torch.randn(N, K, device="cuda").bfloat16()
        dw = name.startswith("dw")
        c = torch.empty(M, N, device="cuda", dtype=torch.float32 if dw else torch.bfloat16)
        if dw:
            td = bench(lambda: ops.gemm_nt(a, b, out_f32=c, accumulate=True))
            t0 = bench(lambda: ops.gemm_nt_variant(0, a, b, out_f32=c, accumulate=True))
            t4 = bench(lambda: ops.gemm_nt_variant(4, a, b, out_f32=c, accumulate=True))
        else:
            td = bench(lambda: ops.gemm_nt(a, b, out=c))
            t0 = bench(lambda: ops.gemm_nt_variant(0, a, b, out=c))
            t4 = bench(lambda: ops.gemm_nt_variant(4, a, b, out=c))
        tb = bench(lambda: torch.matmul(a, b.t()))
        f = 2.0 * M * N * K / 1e12
        for k_, t in (("disp", td), ("v0", t0), ("v4", t4), ("blas", tb)):
            tot[k_] += t
        print(f"T={T} {name:8s} {M:6d}x{N:6d}x{K:6d}: dispatched {f / td:6.0f}  v0 {f / t0:6.0f}  v4 {f / t4:6.0f}  hipblaslt {f / tb:6.0f} TF   ({td * 1e6:7.0f} us)", flush=True)
        del a, b, c
    print(f"T={T} layer total: dispatched {tot['disp'] * 1e3:.2f} ms, v0 {tot['v0'] * 1e3:.2f}, v4 {tot['v4'] * 1e3:.2f}, hipblaslt {tot['blas'] * 1e3:.2f}")
