"""Run ONE GEMM variant a few times (for rocprofv3 --pmc passes): python tools/gemm_one.py <variant | lib> M N K"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatialthinker_amd import ops
v = sys.argv[1]
M, N, K = (int(x) for x in sys.argv[2:5])
a = torch.randn(M, K, device="cuda").bfloat16(); b = torch.randn(N, K, device="cuda").bfloat16()
c = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
for _ in range(5):
    if v == "lib":
        torch.matmul(a, b.t(), out=c)
    else:
        ops.gemm_nt_variant(int(v), a, b, out=c)
torch.cuda.synchronize()
