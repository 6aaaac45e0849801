"""One shape, tail split on or off, a few launches (for rocprofv3 --kernel-trace)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatialthinker_amd import ops
M, N, K, on = (int(x) for x in sys.argv[1:5])
a = torch.randn(M, K, device="cuda").bfloat16(); b = torch.randn(N, K, device="cuda").bfloat16()
c = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
ops.gemm_tail_split(bool(on))
for _ in range(6): ops.gemm_nt(a, b, out=c)
torch.cuda.synchronize()
