"""A/B of the tail split of the 256x256-tile GEMMs (st_gemm_set_workspace) on the 7B training shapes + correctness."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatialthinker_amd import ops

def bench(fn, iters=8):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3

T = int(sys.argv[1]) if len(sys.argv) > 1 else 10400
shapes = {"qkv": (T, 4608, 3584), "o": (T, 3584, 3584), "down": (T, 3584, 18944), "dx_gu": (T, 3584, 37888), "dx_down": (T, 18944, 3584),
          "dW_down": (3584, 18944, T // 64 * 64), "dW_gu": (37888, 3584, T // 64 * 64), "dW_o": (3584, 3584, T // 64 * 64), "lmhead": (2048, 152064, 3584)}
for name, (M, N, K) in shapes.items():
    a = torch.randn(M, K, device="cuda").bfloat16(); b = torch.randn(N, K, device="cuda").bfloat16()
    bias = torch.randn(N, device="cuda").bfloat16(); res = torch.randn(M, N, device="cuda").bfloat16()
    ref = torch.matmul(a, b.t()).float()
    out = {}
    for on in (False, True):
        ops.gemm_tail_split(on)
        c = ops.gemm_nt(a, b)
        c2 = ops.gemm_nt(a, b, bias=bias, residual=res)
        f = torch.ones(M, N, device="cuda"); ops.gemm_nt(a, b, out_f32=f, accumulate=True)
        t = min(bench(lambda: ops.gemm_nt(a, b, out=c)) for _ in range(2))
        out[on] = (c, c2, f, t)
    sc = ref.abs().max().item()
    e_ref = (out[True][0].float() - ref).abs().max().item() / sc
    e_ab = max((out[True][i].float() - out[False][i].float()).abs().max().item() for i in range(3)) / sc
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    print(f"{name:8s} {M}x{N}x{K} tiles {tiles} ({tiles / 256:.2f} rounds): off {2.0*M*N*K/out[False][3]/1e12:6.0f} TF  on {2.0*M*N*K/out[True][3]/1e12:6.0f} TF"
          f"  ({out[False][3]/out[True][3]:.2f}x)  err vs torch {e_ref:.1e}  on-vs-off {e_ab:.1e}", flush=True)
