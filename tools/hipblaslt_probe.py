"""What does the vendor library run at the training GEMM shapes?  torch.matmul (hipBLASLt) at the 7B layer's forward shapes, a few
launches each, meant to sit under `rocprofv3 --kernel-trace --stats` (kernel names carry the Tensile configuration: macro tile, MFMA
shape, LDS/prefetch scheme) and under PMC passes (MFMA busy, clock, LDS and VMEM instruction counts) — the yardstick for st_gemm_nt.
    python tools/hipblaslt_probe.py [ours|lib|both] [T]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
which = sys.argv[1] if len(sys.argv) > 1 else "both"
T = int(sys.argv[2]) if len(sys.argv) > 2 else 10496
SHAPES = [("qkv", 4608, 3584), ("gate_up", 37888, 3584), ("down", 3584, 18944), ("o", 3584, 3584)]
from spatialthinker_amd import ops  # noqa: E402

g = torch.Generator(device="cuda").manual_seed(0)
for name, N, K in SHAPES:
    a = (torch.randn(T, K, device="cuda", generator=g) * 0.5).bfloat16()
    w = (torch.randn(N, K, device="cuda", generator=g) * 0.05).bfloat16()
    out = torch.empty(T, N, device="cuda", dtype=torch.bfloat16)
    for fn, tag in ((lambda: torch.matmul(a, w.t(), out=out), "lib"), (lambda: ops.gemm_nt(a, w, out=out), "ours")):
        if which not in (tag, "both"):
            continue
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print(f"{name:8s} {tag:5s} T={T} N={N} K={K}: {ms:.3f} ms  {2.0 * T * N * K / ms / 1e9:.0f} TF/s", flush=True)
    del a, w, out
