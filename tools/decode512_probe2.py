"""What a tail split would buy the 257..512-row gate/up GEMM: the asm4 tile at one exact round (I = 16384: 256 tiles), and the plain
product at the real width with / without the tail split (296 tiles).      python tools/decode512_probe2.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatialthinker_amd import ops  # noqa: E402
from decode_gemm_tune import timeit  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    K = 3584
    for M in (384, 512):
        a = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
        cnt = [0]

        def nextw(ws):
            cnt[0] += 1
            return ws[cnt[0] % len(ws)]
        for I in (16384, 18944):
            wg = [(torch.randn(2 * I, K, device=dev) * 0.05).to(torch.bfloat16) for _ in range(8)]
            t1 = timeit(lambda: ops.gemm_swiglu(a, nextw(wg), want_gu=False))
            o = torch.empty(M, 2 * I, dtype=torch.bfloat16, device=dev)
            ops.gemm_tail_split(True)
            t2 = timeit(lambda: ops.gemm_nt(a, nextw(wg), out=o))
            ops.gemm_tail_split(False)
            t3 = timeit(lambda: ops.gemm_nt(a, nextw(wg), out=o))
            ops.gemm_tail_split(True)
            fl = 2 * M * 2 * I * K
            print(f"M={M} I={I} tiles={2 * (-(-2 * I // 256))}: swiglu(asm4) {t1 * 1e6:6.1f}us ({fl / t1 / 1e12:5.0f} TF) | plain tail-split {t2 * 1e6:6.1f}us ({fl / t2 / 1e12:5.0f} TF) | plain no split {t3 * 1e6:6.1f}us", flush=True)
            del wg, o


if __name__ == "__main__":
    main()
