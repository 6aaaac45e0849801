"""Per-workgroup time trace of GEMM variant 40 (debug variant 47): when does each tile start, reach its epilogue, finish?  Answers
whether the 256 CUs run their epilogues (the stores) at the same moments.   python tools/tile_trace.py [T] [N]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatialthinker_amd import ops
T = int(sys.argv[1]) if len(sys.argv) > 1 else 28672
N = int(sys.argv[2]) if len(sys.argv) > 2 else 37888
K = 3584
a = (torch.randn(T, K, device="cuda") * 0.1).bfloat16(); w = (torch.randn(N, K, device="cuda") * 0.1).bfloat16()
out = torch.empty(T, N, dtype=torch.bfloat16, device="cuda")
ws = ops._gemm_workspace(torch.device("cuda"))
for _ in range(3): ops.gemm_nt_variant(47, a, w, out=out)
torch.cuda.synchronize()
nt = ((T + 255) // 256) * ((N + 255) // 256)
tr = ws[: nt * 32].view(torch.int64).view(nt, 4).cpu().numpy()
t0 = tr[:, 0].min()
start, epi, end = (tr[:, 0] - t0) / 100.0, (tr[:, 1] - t0) / 100.0, (tr[:, 2] - t0) / 100.0     # us (100 MHz)
print(f"{nt} tiles, kernel span {end.max():.1f} us; K loop {np.mean(epi - start):.2f} us (p10 {np.percentile(epi - start, 10):.2f}, p90 {np.percentile(epi - start, 90):.2f}); "
      f"epilogue {np.mean(end - epi):.2f} us (p10 {np.percentile(end - epi, 10):.2f}, p50 {np.percentile(end - epi, 50):.2f}, p90 {np.percentile(end - epi, 90):.2f}, max {np.max(end - epi):.2f})")
# concurrency: how many workgroups are in their epilogue at each 0.5-us instant
grid = np.arange(0, end.max(), 0.5)
conc = np.array([np.sum((epi <= g) & (end > g)) for g in grid])
print("epilogues in flight at a time: mean %.1f, p50 %d, p90 %d, max %d; fraction of time with > 64 in flight: %.2f" % (conc.mean(), np.percentile(conc, 50), np.percentile(conc, 90), conc.max(), np.mean(conc > 64)))
# epilogue duration as a function of concurrency at its start
c_at = np.array([np.sum((epi <= e) & (end > e)) for e in epi[:: max(1, nt // 2000)]])
d_at = (end - epi)[:: max(1, nt // 2000)]
for lo, hi in ((0, 8), (8, 32), (32, 64), (64, 128), (128, 257)):
    m = (c_at >= lo) & (c_at < hi)
    if m.any(): print(f"  epilogues starting with {lo:3d}..{hi - 1:3d} others in flight: n={int(m.sum()):5d}  mean duration {d_at[m].mean():6.2f} us")
# first rounds: spread of epilogue start times
order = np.argsort(start)
for r in (0, 1, 2, 5, 10, 30, 60):
    sel = order[r * 256:(r + 1) * 256]
    if len(sel) == 256: print(f"  dispatch round {r:2d}: starts span {start[sel].max() - start[sel].min():6.2f} us (std {start[sel].std():5.2f}); epilogue starts std {epi[sel].std():5.2f} us")
hw = tr[:, 3]
print("distinct (xcc, hw_id) pairs:", len(np.unique(hw)))
