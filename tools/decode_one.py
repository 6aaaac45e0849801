"""Run ONE decode-shaped GEMM variant over rotating weight copies (for rocprofv3 --pmc passes): decode_one.py variant splits M N K"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatialthinker_amd import ops
from spatialthinker_amd.lib import lib
v, sp, M, N, K = (int(x) for x in sys.argv[1:6])
a = torch.randn(M, K, device="cuda").bfloat16()
ws = [(torch.randn(N, K, device="cuda") * 0.05).bfloat16() for _ in range(8)]
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
scratch = torch.empty(64 << 20, dtype=torch.float32, device="cuda")
for i in range(16):
    w = ws[i % 8]
    lib().st_gemm_nt_decode_variant(v, sp, a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), None, None, 0, out.data_ptr(), out.stride(0),
                                    scratch.data_ptr(), scratch.numel(), M, N, K, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
