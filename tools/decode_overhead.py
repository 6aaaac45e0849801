"""Fixed cost of a decode GEMM tile launch: the qkv / o projections at 512 rows with K swept (split-K off), replayed from a hipGraph;
time(K) = overhead + K/64 * t_ktile.  Also the slab (fp32 split-K partials) form the decode loop uses."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatialthinker_amd import ops
from spatialthinker_amd.lib import lib
from tools.decode_gemm_tune import timeit
_p = ops._p; _s = ops._s
dev = torch.device("cuda")
scratch = torch.empty(64 << 20, dtype=torch.float32, device=dev)
for M in (512, 256):
    for name, N in (("qkv", 4608), ("o", 3584)):
        for v in ((16, 13) if M > 256 else (14, 13)):
            res = {}
            for K in (256, 1024, 3584):
                a = (torch.randn(M, K, device=dev) * 0.5).bfloat16()
                ws = [(torch.randn(N, K, device=dev) * 0.05).bfloat16() for _ in range(8)]
                out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
                i = [0]
                def fn():
                    w = ws[i[0] % 8]; i[0] += 1
                    lib().st_gemm_nt_decode_variant(v, 1, _p(a), a.stride(0), _p(w), w.stride(0), None, None, 0, _p(out), out.stride(0), _p(scratch), scratch.numel(), M, N, K, _s())
                res[K] = timeit(fn) * 1e6
            kt = (res[3584] - res[256]) / 52.0
            print(f"M={M} {name} variant {v}: K=256 {res[256]:6.1f} us, K=1024 {res[1024]:6.1f}, K=3584 {res[3584]:6.1f}  -> {kt:.2f} us per K-tile, fixed {res[256] - 4 * kt:5.1f} us", flush=True)
# an empty kernel launch inside a graph, for scale
x = torch.zeros(256, device=dev)
print(f"graph-replayed tiny elementwise kernel: {timeit(lambda: x.add_(1.0)) * 1e6:.1f} us")
