"""Run the decode gate/up GEMM with the SwiGLU epilogue (256x160 3-slot tile) over rotating weight copies, for rocprofv3 --pmc
passes: decode_swiglu_one.py M I K"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatialthinker_amd import ops
M, I, K = (int(x) for x in sys.argv[1:4])
a = torch.randn(M, K, device="cuda").bfloat16()
ws = [(torch.randn(2 * I, K, device="cuda") * 0.05).bfloat16() for _ in range(8)]
out = torch.empty(M, I, device="cuda", dtype=torch.bfloat16)
for i in range(16):
    ops.gemm_swiglu_decode(a, ws[i % 8], out=out)
torch.cuda.synchronize()
