"""Decode gate/up + SwiGLU GEMM: every tile variant of st_gemm_swiglu_decode_variant vs the default plan, cold weights, graph replay.

    python tools/decode_swiglu_tune.py [M ...]        (7B: I = 18944, K = 3584)
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatialthinker_amd import ops  # noqa: E402
from spatialthinker_amd.lib import lib  # noqa: E402
from decode_gemm_tune import timeit  # noqa: E402


def main():
    Ms = [int(a) for a in sys.argv[1:]] or [64, 128, 256, 512]
    I, K = 18944, 3584
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    ws = [(torch.randn(2 * I, K, device=dev) * 0.05).to(torch.bfloat16) for _ in range(8)]
    for M in Ms:
        a = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
        out = torch.empty(M, I, dtype=torch.bfloat16, device=dev)
        ref = ops.gemm_swiglu_decode(a, ws[0]).float()
        cnt = [0]

        def nextw():
            cnt[0] += 1
            return ws[cnt[0] % len(ws)]
        t0 = timeit(lambda: ops.gemm_swiglu_decode(a, nextw(), out=out))
        line = f"M={M:3d}: default {t0 * 1e6:6.1f}us ({2 * M * 2 * I * K / t0 / 1e12:5.0f} TF, {2 * I * K * 2 / t0 / 1e12:4.2f} TB/s) |"
        for v in range(1, 8):
            if (v == 6 and M > 128) or (v == 7 and M > 64):
                continue

            def run(w=None):
                w = nextw() if w is None else w
                lib().st_gemm_swiglu_decode_variant(v, a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), out.data_ptr(), out.stride(0), M, I, K,
                                                    torch.cuda.current_stream().cuda_stream)
            out.zero_()
            run(ws[0])
            torch.cuda.synchronize()
            err = (out.float() - ref).abs().max().item()
            t = timeit(run)
            line += f" v{v}:{t * 1e6:.0f}" + ("" if err == 0.0 else f"(err {err:.1e})")
        print(line, flush=True)


if __name__ == "__main__":
    main()
