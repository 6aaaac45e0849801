"""Run ONE GEMM variant a few times (for rocprofv3 --pmc passes)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatialthinker_amd import ops
v, M, N, K = (int(x) for x in sys.argv[1:5])
a = torch.randn(M, K, device="cuda").bfloat16(); b = torch.randn(N, K, device="cuda").bfloat16()
c = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
for _ in range(5): ops.gemm_nt_variant(v, a, b, out=c)
torch.cuda.synchronize()
