"""Where does a wave of the tiled GEMM spend its K loop?  Needs a library built with -DST_GEMM_TRACE (tools/gemm_phase_trace.sh): wave 0
of every workgroup sums the shader-clock cycles between fixed points of the loop (see the TR_POINT comments in gemm_tiles.hip).

    ST_LIB=variants/trace.so python tools/gemm_phase_trace.py
"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spatialthinker_amd.lib as _lib  # noqa: E402
_lib.LIB_PATH = os.path.abspath(os.environ["ST_LIB"])
from spatialthinker_amd import ops  # noqa: E402
from spatialthinker_amd.lib import lib  # noqa: E402

NAMES_RING = ["wait copies", "barrier", "frag reads k0 (exposed)", "DMA issue + MFMA k0 + reads k1", "MFMA k1"]
NAMES_MID = ["phase A issue (MFMA k0 + reads k1)", "wait copies", "frag reads returned", "barrier", "phase B (MFMA k1 + reads + DMA issue)"]


def report(title, buf, n_wg, nk, names, flops_per_ktile_wave, nw=8):
    torch.cuda.synchronize()
    t = buf[:n_wg * nw * 8].view(n_wg, nw, 8).double().mean(0).cpu().numpy() / nk        # [wave][point]
    tot = t[:, :len(names)].sum(1)
    print(f"{title}: {tot.mean():7.0f} cycles per K-tile per wave (MFMA-only time of the SIMD's two waves: {2 * flops_per_ktile_wave / 1024:.0f})")
    print("    " + " " * 42 + "".join(f"   wave{w}" for w in range(nw)))
    for i, nm in enumerate(names):
        print(f"    {nm:42s}" + "".join(f"{t[w, i]:8.0f}" for w in range(nw)))


def main():
    dev = torch.device("cuda:0")
    so = ctypes.CDLL(_lib.LIB_PATH)
    buf = torch.zeros(1 << 20, dtype=torch.int64, device=dev)
    so.st_gemm_trace_set.argtypes = [ctypes.c_void_p]
    assert so.st_gemm_trace_set(buf.data_ptr()) == 0
    torch.manual_seed(0)
    I, K = 18944, 3584
    ws = [(torch.randn(2 * I, K, device=dev) * 0.05).to(torch.bfloat16) for _ in range(6)]
    for M in (256, 512):
        a = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
        out = torch.empty(M, I, dtype=torch.bfloat16, device=dev)
        for v, (bm, bn, names) in {1: (256, 160, NAMES_RING), 5: (256, 192, NAMES_MID), 4: (256, 256, NAMES_MID)}.items():
            for i in range(5):
                buf.zero_()
                w = ws[i % len(ws)]
                lib().st_gemm_swiglu_decode_variant(v, a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), out.data_ptr(), out.stride(0), M, I, K,
                                                    torch.cuda.current_stream().cuda_stream)
            n_wg = -(-M // bm) * -(-I // (bn // 2))
            report(f"decode gate/up + SwiGLU M={M} tile {bm}x{bn} (variant {v}, {n_wg} workgroups)", buf, n_wg, K // 64, names,
                   (bm // 4) * (bn // 2) * 64 * 2)
    # training tile: qkv forward at T = 10496
    T, N = 10496, 4608
    a = (torch.randn(T, K, device=dev) * 0.5).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
    out = torch.empty(T, N, dtype=torch.bfloat16, device=dev)
    ops.gemm_tail_split(False)
    for i in range(3):
        buf.zero_()
        ops.gemm_nt(a, w, out=out)
    n_wg = -(-T // 256) * -(-N // 256)
    report(f"training tile 256x256 (mid-tile barrier), {T}x{N}x{K}, {n_wg} workgroups", buf, n_wg, K // 64, NAMES_MID, 64 * 128 * 64 * 2)
    for i in range(3):
        buf.zero_()
        ops.gemm_nt_variant(31, a, w, out=out)
    torch.cuda.synchronize()
    report("ping-pong schedule (variant 31)", buf, n_wg, K // 64, ["M0 issue", "barrier", "R0 reads (+ wait copies, group 1)", "barrier",
           "M1 issue (+ wait copies, group 0)", "barrier", "R1 DMA issue + reads", "barrier"], 64 * 128 * 64 * 2)


if __name__ == "__main__":
    main()
