"""Fixed per-tile cost (prologue + epilogue + dispatch) of the 256x256 GEMM: time at K = 64..512 on exactly 4 rounds of 256 tiles
(M = N = 8192), per epilogue kind; the intercept of the line through the points is the per-tile overhead."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatialthinker_amd import ops
def bench(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3   # us
M = N = 8192
v = int(sys.argv[1]) if len(sys.argv) > 1 else 6
bias = torch.randn(N, device="cuda").bfloat16(); res = torch.randn(M, N, device="cuda").bfloat16()
c = torch.empty(M, N, device="cuda", dtype=torch.bfloat16); f = torch.zeros(M, N, device="cuda")
for K in (64, 128, 256, 512, 1024):
    a = torch.randn(M, K, device="cuda").bfloat16(); b = torch.randn(N, K, device="cuda").bfloat16()
    t0 = bench(lambda: ops.gemm_nt_variant(v, a, b, out=c))
    t1 = bench(lambda: ops.gemm_nt_variant(v, a, b, out=c, bias=bias, residual=res))
    t2 = bench(lambda: ops.gemm_nt_variant(v, a, b, out_f32=f, accumulate=True))
    print(f"v{v} K={K:5d} ({K // 64:2d} K-tiles) 4 rounds: bf16 {t0:7.1f} us  bias+res {t1:7.1f} us  f32 accumulate {t2:7.1f} us   per round {t0 / 4:6.1f} / {t1 / 4:6.1f} / {t2 / 4:6.1f}", flush=True)
