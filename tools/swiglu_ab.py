"""gate/up forward at the fused-pass token count: fused SwiGLU epilogue (with / without the kept gate|up) vs plain GEMM + st_swiglu_fwd."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatialthinker_amd import ops
T, H, I = int(sys.argv[1]) if len(sys.argv) > 1 else 21504, 3584, 18944
x = (torch.randn(T, H, device="cuda") * 0.1).bfloat16(); w = (torch.randn(2 * I, H, device="cuda") * 0.1).bfloat16()
ops._gemm_workspace(torch.device("cuda"))
def bench(fn, iters=6):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
gu_buf = torch.empty(T, 2 * I, dtype=torch.bfloat16, device="cuda")
for v in (40, 23):
    ops.gemm_select(v)
    t1 = bench(lambda: ops.gemm_swiglu(x, w, want_gu=True))
    t2 = bench(lambda: ops.gemm_swiglu(x, w, want_gu=False))
    t3 = bench(lambda: ops.gemm_nt(x, w, out=gu_buf))
    t4 = bench(lambda: ops.swiglu_fwd(gu_buf))
    fl = 2.0 * T * 2 * I * H / 1e9
    print(f"v{v}: fused+gu {t1:.3f} ms ({fl/t1:.0f} TF/s)  fused m only {t2:.3f} ({fl/t2:.0f})  plain gemm {t3:.3f} ({fl/t3:.0f})  + swiglu kernel {t4:.3f} -> unfused {t3+t4:.3f} ms")
ops.gemm_select(40)
