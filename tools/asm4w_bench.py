"""The 4-wave bf16 tile on the 32-cycle MFMA (gemm_asm4w.hip, variants 48..51 = K-tile schedules 0..3) against the production tile (40):
results vs the fp32 product, then throughput on the layer's forward shapes.   python tools/asm4w_bench.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatialthinker_amd import ops
torch.manual_seed(0)
ops._gemm_workspace(torch.device("cuda"))
def timeit(fn, iters=20):
    for _ in range(40): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
print("== results")
for (M, N, K, hb, hr) in ((300, 520, 192, True, True), (256, 256, 64, False, False), (512, 768, 128, True, False), (1024, 512, 3584, False, True), (2048, 3584, 18944, False, True)):
    a = torch.randn(M, K, device="cuda").bfloat16(); b = (torch.randn(N, K, device="cuda") * (1 + torch.arange(N, device="cuda")[:, None] / N)).bfloat16()
    bias = torch.randn(N, device="cuda").bfloat16() if hb else None
    res = torch.randn(M, N, device="cuda").bfloat16() if hr else None
    want = a.float() @ b.float().t() + (bias.float() if hb else 0) + (res.float() if hr else 0)
    r40 = ops.gemm_nt_variant(40, a, b, bias=bias, residual=res)
    line = f"{M}x{N}x{K} bias={hb} res={hr}: v40 err {float((r40.float() - want).abs().max()):.4f}"
    for v in (48, 49, 50, 51):
        if v != 48 and (hb or hr): continue
        r = ops.gemm_nt_variant(v, a, b, bias=bias, residual=res)
        line += f" | v{v} err {float((r.float() - want).abs().max()):.4f} identical to v40: {bool(torch.equal(r, r40))}"
    print(line, "(scale", float(want.abs().max()), ")", flush=True)
print("== throughput")
H, QKV, I2, I = 3584, 4608, 37888, 18944
for name, M, N, K in [("qkv", 16384, QKV, H), ("o", 16384, H, H), ("gateup", 16384, I2, H), ("down", 16384, H, I), ("o 10496", 10496, H, H), ("4096x4096x8192", 4096, 4096, 8192),
                      ("7 rounds K=3584", 28672, 4096, 3584)]:
    a = (torch.randn(M, K, device="cuda") * 0.5).bfloat16(); b = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    c = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    fl = 2.0 * M * N * K / 1e6
    line = f"{name:16s}"
    for v in (40, 48, 49, 50, 51, 40, 48):
        t = timeit(lambda: ops.gemm_nt_variant(v, a, b, out=c))
        line += f"  v{v} {t:7.1f} us {fl / t:5.0f} TF"
    print(line, flush=True)
