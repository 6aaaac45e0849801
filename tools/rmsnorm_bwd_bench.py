"""RMSNorm backward: the two-kernel form vs the one-pass kernel at the update pass's shape.   python tools/rmsnorm_bwd_bench.py [T] [H]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spatialthinker_amd.ops as O
T = int(sys.argv[1]) if len(sys.argv) > 1 else 20864
H = int(sys.argv[2]) if len(sys.argv) > 2 else 3584
x, dy, dres = (torch.randn(T, H, device="cuda").bfloat16() for _ in range(3))
w = torch.ones(H, device="cuda").bfloat16()
_, rstd = O.rmsnorm_fwd(x, w, 1e-6)
dw = torch.zeros(H, dtype=torch.float32, device="cuda")
for fused in (False, True, False, True):
    O.RMSNORM_BWD_FUSED = fused
    for _ in range(3):
        O.rmsnorm_bwd(x, w, rstd, dy, dres=dres, dw_accum=dw)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 50
    for _ in range(n):
        O.rmsnorm_bwd(x, w, rstd, dy, dres=dres, dw_accum=dw)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    gb = 4 * T * H * 2 / 1e9
    print(f"T={T} H={H} fused={fused}: {dt * 1e6:.1f} us per call, {gb / dt / 1e3:.2f} TB/s of the 4 algorithmic passes (x, dy, dres in, dx out)")
