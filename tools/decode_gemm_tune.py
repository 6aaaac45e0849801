"""Decode-shaped GEMM sweep on the 7B LM shapes: tile variant x split-K vs the current st_gemm_nt_skinny.

    python tools/decode_gemm_tune.py [M ...]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatialthinker_amd import ops  # noqa: E402
from spatialthinker_amd.lib import lib  # noqa: E402

SHAPES = [("qkv", 4608, 3584), ("o", 3584, 3584), ("down", 3584, 18944)] if os.environ.get("ST_TUNE_NARROW") else [("qkv", 4608, 3584), ("o", 3584, 3584), ("gateup", 37888, 3584), ("down", 3584, 18944), ("lmhead", 152064, 3584)]
VARIANTS = {64: [10, 11, 12, 21, 13, 14], 128: [13, 14, 19, 20, 16], 256: [16, 18, 28, 13, 14]}
SPLITS = (1, 2, 3, 4, 5, 6, 7, 8, 9, 12, 16, 18)


def timeit(fn, n=32):
    """GPU time per call with the launches replayed from a hipGraph (no host launch overhead, like the decode loop)."""
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        with torch.cuda.graph(g, stream=st):
            for _ in range(n):
                fn()
    g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (3 * n)


def main():
    Ms = [int(a) for a in sys.argv[1:]] or [64, 256]
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    scratch = torch.empty(64 << 20, dtype=torch.float32, device=dev)
    for M in Ms:
        bm = 64 if M <= 64 else (128 if M <= 128 else 256)
        wide = M > 256                                      # 257..512-row decode batches: two 256-row tiles per projection (ops.gemm_nt(decode=True))
        for name, N, K in SHAPES:
            a = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
            # the decode loop streams ~15 GB of weights per step: every layer's W arrives cold.  Rotate over enough copies of
            # W (>= 1.5 GB footprint, beyond the 256 MB MALL) so the timing is HBM-bound like the real loop, not cache-resident.
            ncopy = max(2, min(32, int(1.5e9 // (N * K * 2)) + 1))
            ws = [(torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16) for _ in range(ncopy)]
            w = ws[0]
            bias = torch.randn(N, device=dev).to(torch.bfloat16)
            ref = (a.float() @ w.float().t() + bias.float())
            out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
            cnt = [0]

            def nextw():
                cnt[0] += 1
                return ws[cnt[0] % ncopy]
            t0 = timeit(lambda: ops.gemm_nt(a, nextw(), bias=bias, out=out, decode=wide))
            e0 = (out.float() - ref).abs().max().item()
            gb = N * K * 2 / 1e9
            line = f"M={M:3d} {name:7s} N={N:6d} K={K:5d}: default plan {t0 * 1e6:7.1f}us ({gb / t0 / 1e3:4.2f} TB/s, err {e0:.1e}) |"
            best = (t0, "default")
            for v in [0] + VARIANTS[bm]:
                for sp in ((1,) if v == 0 else SPLITS):
                    if sp > 1 and sp * M * N > scratch.numel():
                        continue
                    if sp > 1 and N > 40000:
                        continue

                    def run():
                        w = nextw()
                        lib().st_gemm_nt_decode_variant(v, sp, a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), bias.data_ptr(), None, 0,
                                                        out.data_ptr(), out.stride(0), scratch.data_ptr(), scratch.numel(), M, N, K,
                                                        torch.cuda.current_stream().cuda_stream)
                    try:
                        out.zero_()
                        cnt[0] = -1
                        run()
                        torch.cuda.synchronize()
                    except Exception as ex:  # noqa: BLE001
                        line += f" v{v}s{sp}:ERR"
                        continue
                    err = (out.float() - ref).abs().max().item()
                    t = timeit(run)
                    if err > 0.1:
                        line += f" v{v}s{sp}:BAD({err:.1e})"
                        continue
                    if t < best[0]:
                        best = (t, f"v{v}s{sp}")
                    line += f" v{v}s{sp}:{t * 1e6:.0f}"
            print(line)
            print(f"    best {best[1]} {best[0] * 1e6:.1f}us = {gb / best[0] / 1e3:.2f} TB/s, {2 * M * N * K / best[0] / 1e12:.0f} TF", flush=True)


if __name__ == "__main__":
    main()
