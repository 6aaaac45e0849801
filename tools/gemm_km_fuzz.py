"""Random-shape check of the contraction-major forms of the 4-wave tile (st_gemm_tn, st_gemm_nn at M > 256) against fp32 torch."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatialthinker_amd import ops
rs = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
ops.gemm_select(40)
bad = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    M, N, K = 8 * rs.randint(1, 140), 8 * rs.randint(1, 140), 64 * rs.randint(1, 12)
    split = bool(rs.randint(2)); ops.gemm_tail_split(split)
    a_km = torch.from_numpy(rs.standard_normal((K, M)).astype(np.float32)).bfloat16().cuda()
    b_kn = torch.from_numpy(rs.standard_normal((K, N)).astype(np.float32)).bfloat16().cuda()
    want = a_km.float().t() @ b_kn.float()
    out = torch.full((M, N), 0.25, dtype=torch.float32, device="cuda")
    ops.gemm_tn(a_km, b_kn, out, accumulate=True)
    e1 = float((out - want - 0.25).abs().max() / (want.abs().max() + 1e-6))
    ops.gemm_tn(a_km, b_kn, out, accumulate=False)
    e2 = float((out - want).abs().max() / (want.abs().max() + 1e-6))
    e3 = 0.0
    if M > 256:
        a_mk = a_km.t().contiguous()
        o = ops.gemm_nn(a_mk, b_kn)
        e3 = float((o.float() - want).abs().max() / (want.abs().max() + 1e-6))
    ok = e1 < 2e-5 and e2 < 2e-5 and e3 < 2 ** -7
    bad += not ok
    if not ok or it < 3: print(f"M={M} N={N} K={K} split={split}: tn+= {e1:.2e} tn {e2:.2e} nn {e3:.2e} {'OK' if ok else 'FAIL'}", flush=True)
ops.gemm_tail_split(True)
print("failures:", bad)
