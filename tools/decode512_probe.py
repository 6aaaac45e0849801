"""257..512-row decode GEMMs: the decode tile plans vs the training-tile entries (st_gemm_swiglu / st_gemm_nt) at the same shapes,
cold weights, graph replay.      python tools/decode512_probe.py [M ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatialthinker_amd import ops  # noqa: E402
from decode_gemm_tune import timeit  # noqa: E402


def main():
    Ms = [int(a) for a in sys.argv[1:]] or [320, 384, 512]
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    I, K = 18944, 3584
    wg = [(torch.randn(2 * I, K, device=dev) * 0.05).to(torch.bfloat16) for _ in range(8)]
    shapes = [("qkv", 4608, 3584), ("o", 3584, 3584), ("down", 3584, 18944)]
    for M in Ms:
        a = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
        out = torch.empty(M, I, dtype=torch.bfloat16, device=dev)
        cnt = [0]

        def nextw(ws):
            cnt[0] += 1
            return ws[cnt[0] % len(ws)]
        ref = ops.gemm_swiglu_decode(a, wg[0]).float()
        _, m = ops.gemm_swiglu(a, wg[0], want_gu=False)
        err = (m.float() - ref).abs().max().item()
        t0 = timeit(lambda: ops.gemm_swiglu_decode(a, nextw(wg), out=out))
        t1 = timeit(lambda: ops.gemm_swiglu(a, nextw(wg), want_gu=False))
        fl = 2 * M * 2 * I * K
        print(f"M={M} gate/up+swiglu: decode plan {t0 * 1e6:6.1f}us ({fl / t0 / 1e12:5.0f} TF) | st_gemm_swiglu {t1 * 1e6:6.1f}us ({fl / t1 / 1e12:5.0f} TF) err {err:.2e}", flush=True)
        for name, N, Kk in shapes:
            aa = (torch.randn(M, Kk, device=dev) * 0.5).to(torch.bfloat16)
            ncopy = max(2, min(32, int(1.5e9 // (N * Kk * 2)) + 1))
            ws = [(torch.randn(N, Kk, device=dev) * 0.05).to(torch.bfloat16) for _ in range(ncopy)]
            o = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
            t0 = timeit(lambda: ops.gemm_nt(aa, nextw(ws), out=o, decode=True))
            t1 = timeit(lambda: ops.gemm_nt(aa, nextw(ws), out=o, decode=False))
            fl = 2 * M * N * Kk
            print(f"M={M} {name:5s}: decode plan {ops.decode_plan(M, N, Kk)} {t0 * 1e6:6.1f}us ({fl / t0 / 1e12:5.0f} TF) | st_gemm_nt {t1 * 1e6:6.1f}us ({fl / t1 / 1e12:5.0f} TF)", flush=True)


if __name__ == "__main__":
    main()
