"""Where the 4-wave fp8 tile's K loop goes, by elimination: python tools/mx4_ksweep.py [M N K]
Times st_gemm_mxfp8_nt on the real kernel and on its timing-experiment instantiations (results wrong by construction): no LDS-DMA in the
loop / no barriers / no fragment + scale reads / MFMAs only.  One full round of tiles (256 CUs) unless a shape is given."""
import os, sys, torch
os.environ["ST_FP8_TIMING_EXPERIMENTS"] = "1"      # the wrong-result instantiations are refused otherwise
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatialthinker_amd import ops
M, N, K = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (4096, 4096, 8192)
a = torch.randn(M, K, device="cuda").bfloat16(); b = torch.randn(N, K, device="cuda").bfloat16()
c = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
aq, sa = ops.mxfp8_quantize(a); bq, sb = ops.mxfp8_quantize(b)
def timeit(fn, iters=20):
    for _ in range(40): fn()              # the first launches after a pause run at a lower clock: warm up for ~5-50 ms
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
tiles = -(-M // 256) * -(-N // 256)
nk = K // 128
print(f"{M}x{N}x{K}: {tiles} tiles ({tiles / 256:.2f} rounds), {nk} K-tiles")
modes = ((4, "real kernel"), (41, "no LDS-DMA in the loop"), (42, "no barriers"), (43, "no fragment / scale reads"), (44, "MFMAs only"))
if os.environ.get("MX4_SCHEDULES"):
    modes = tuple((50 + v, f"schedule {v}") for v in (int(x) for x in os.environ["MX4_SCHEDULES"].split(",")))          # e.g. MX4_SCHEDULES=6,0,2,6
for mode, what in modes:
    ops.gemm_mxfp8_select(mode)
    t = timeit(lambda: ops.gemm_mxfp8_nt(aq, sa, bq, sb, out=c))
    rounds = -(-tiles // 256)
    print(f"  {what:28s} {t:8.1f} us = {2.0 * M * N * K / t / 1e6:6.0f} TF/s; {t / rounds / nk * 1e3:7.1f} ns per K-tile and round", flush=True)
ops.gemm_mxfp8_select(4)
