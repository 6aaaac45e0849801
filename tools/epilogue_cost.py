"""Per-tile cost of the variant-40 epilogue by elimination, at the gate/up forward shape: the full kernel, the kernel without its global
stores (LDS staging + read-out kept), and the kernel without any epilogue.  Same launch, alternating order, sustained."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatialthinker_amd import ops
T = int(sys.argv[1]) if len(sys.argv) > 1 else 28672
N = int(sys.argv[2]) if len(sys.argv) > 2 else 37888
K = 3584
a = (torch.randn(T, K, device="cuda") * 0.1).bfloat16(); w = (torch.randn(N, K, device="cuda") * 0.1).bfloat16()
out = torch.empty(T, N, dtype=torch.bfloat16, device="cuda")
ops._gemm_workspace(torch.device("cuda"))
tiles_per_cu = max(1.0, ((T + 255) // 256) * ((N + 255) // 256) / 256.0)
def bench(v, iters=8 if T * N > 1 << 28 else 200):
    for _ in range(2): ops.gemm_nt_variant(v, a, w, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): ops.gemm_nt_variant(v, a, w, out=out)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for rep in range(2):
    for v, tag in ((40, "full"), (46, "no global stores"), (45, "no epilogue"), (44, "MFMA only")):
        ms = bench(v)
        print(f"v{v} {tag:18s}: {ms:.3f} ms  {ms * 1e3 / tiles_per_cu:.2f} us per tile  ({2.0*T*N*K/ms/1e9:.0f} TF/s)", flush=True)
