#!/bin/bash
# 32-cycle-MFMA bf16 tile: K loop by elimination next to the production tile's (variants 41 no DMA, 43 no reads, 44 MFMAs only)
mkdir -p gpurun_out/r04
python3 - <<'PY' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04/asm4w_elim.txt
import sys, torch
sys.path.insert(0, ".")
from spatialthinker_amd import ops
ops._gemm_workspace(torch.device("cuda"))
def timeit(fn, iters=20):
    for _ in range(40): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for (M, N, K) in ((4096, 4096, 8192), (28672, 4096, 3584), (16384, 3584, 18944)):
    a = (torch.randn(M, K, device="cuda") * 0.5).bfloat16(); b = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    c = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    fl = 2.0 * M * N * K / 1e6
    line = f"{M}x{N}x{K}:"
    for v, what in ((40, "v40"), (41, "v40 no DMA"), (43, "v40 no reads"), (44, "v40 MFMA only"), (49, "w4"), (52, "w4 no DMA"), (53, "w4 no reads"), (54, "w4 MFMA only"), (40, "v40")):
        t = timeit(lambda: ops.gemm_nt_variant(v, a, b, out=c))
        line += f"  {what} {t:7.1f} us {fl / t:5.0f} TF |"
    print(line, flush=True)
PY
