"""Power-steady-state GEMM throughput: each candidate runs back to back for several seconds (the chip settles at the clock its power
budget allows — a few launches between idle gaps over-state what a training step sustains), TF/s over the second half.
    python tools/gemm_sustained.py [seconds] [T]      candidates: st_gemm_nt tile variants 23 / 6 / 8 / 31 and hipBLASLt (torch.matmul)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatialthinker_amd import ops  # noqa: E402

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
T = int(sys.argv[2]) if len(sys.argv) > 2 else 28672
g = torch.Generator(device="cuda").manual_seed(0)
ops._gemm_workspace(torch.device("cuda"))
for name, N, K in (("gate_up", 37888, 3584), ("down", 3584, 18944)):
    a = (torch.randn(T, K, device="cuda", generator=g) * 0.5).bfloat16()
    w = (torch.randn(N, K, device="cuda", generator=g) * 0.05).bfloat16()
    out = torch.empty(T, N, device="cuda", dtype=torch.bfloat16)
    cands = [(f"v{v}", (lambda v=v: ops.gemm_nt_variant(v, a, w, out=out))) for v in (23, 40)] + [("hipblaslt", lambda: torch.matmul(a, w.t(), out=out))]
    for zero in (False, True):
        if zero:
            a.zero_()                                            # the DVFS ceiling: no operand toggling (MI355X_MICROARCH.md, DVFS give-back)
        for tag, fn in cands:
            fn(); torch.cuda.synchronize()
            t0 = time.perf_counter(); n_half = 0; t_half = None; n = 0
            while True:
                for _ in range(10):
                    fn()
                n += 10
                torch.cuda.synchronize()
                now = time.perf_counter()
                if t_half is None and now - t0 >= secs / 2:
                    t_half, n_half = now, n
                if now - t0 >= secs:
                    break
            tf = 2.0 * T * N * K * (n - n_half) / (now - t_half) / 1e12
            print(f"{name:8s} T={T} {'zeros' if zero else 'randn'} {tag:10s}: {tf:7.0f} TF/s sustained over {now - t_half:.1f} s", flush=True)
        if zero:
            break
    del a, w, out
