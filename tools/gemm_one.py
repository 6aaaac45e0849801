"""Run ONE GEMM a few times (for rocprofv3 --pmc passes):
    python tools/gemm_one.py <variant | lib | tn | nn> M N K      tn: out_f32[M,N] += A[K,M]^T B[K,N] (st_gemm_tn), nn: out[M,N] = A[M,K] B[K,N]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatialthinker_amd import ops
v = sys.argv[1]
M, N, K = (int(x) for x in sys.argv[2:5])
rb = lambda *s: torch.randn(*s, device="cuda").bfloat16()
if v == "tn":
    a, b, c = rb(K, M), rb(K, N), torch.zeros(M, N, device="cuda")
    fn = lambda: ops.gemm_tn(a, b, c, accumulate=True)
elif v == "nn":
    a, b, c = rb(M, K), rb(K, N), torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    fn = lambda: ops.gemm_nn(a, b, out=c)
else:
    a, b, c = rb(M, K), rb(N, K), torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    fn = (lambda: torch.matmul(a, b.t(), out=c)) if v == "lib" else (lambda: ops.gemm_nt_variant(int(v), a, b, out=c))
for _ in range(5):
    fn()
torch.cuda.synchronize()
