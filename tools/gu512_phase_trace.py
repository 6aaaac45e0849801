"""Where does a wave of the one-pass 512-row gate/up tile (gemm_swiglu512.hip, ping-pong schedule) spend a K-step?  Needs a library whose
gemm_swiglu512.hip was compiled with -DST_GU512_TRACE (tools/gu512_phase_trace.sh builds variants/trace512.so): every wave sums the
shader-clock cycles between fixed points of its loop.   ST_LIB=variants/trace512.so [ST_GU512_MODE=3] python tools/gu512_phase_trace.py [rows]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spatialthinker_amd.lib as _lib  # noqa: E402
_lib.LIB_PATH = os.path.abspath(os.environ["ST_LIB"])
from spatialthinker_amd.lib import lib  # noqa: E402

NAMES = ["MFMAs of K-step ks (40, issue)", "group 0: wait for its copies of ks + 1", "barrier 1", "copy issues (K-step ks + 3)",
         "fragment reads of ks + 1, returned", "group 1: wait for its copies of ks + 2", "barrier 2"]


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    dev = torch.device("cuda:0")
    so = ctypes.CDLL(_lib.LIB_PATH)
    buf = torch.zeros(1 << 20, dtype=torch.int64, device=dev)
    so.st_gu512_trace_set.argtypes = [ctypes.c_void_p]
    assert so.st_gu512_trace_set(buf.data_ptr()) == 0
    torch.manual_seed(0)
    I, K = int(os.environ.get('ST_TRACE_I', '18944')), 3584
    ws = [(torch.randn(2 * I, K, device=dev) * 0.05).to(torch.bfloat16) for _ in range(int(os.environ.get('ST_TRACE_COPIES', '6')))]
    a = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
    out = torch.empty(M, I, dtype=torch.bfloat16, device=dev)
    def launch(i):
        w = ws[i % len(ws)]
        lib().st_gemm_swiglu_decode_variant(512, a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), out.data_ptr(), out.stride(0), M, I, K,
                                            torch.cuda.current_stream().cuda_stream)
    for i in range(200):                                     # bring the clocks to their loaded state
        launch(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(60):
        launch(i)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 60
    buf.zero_()
    launch(0)
    torch.cuda.synchronize()
    n_wg, nk = -(-I // 80), K // 32
    t = buf[:n_wg * 8 * 8].view(n_wg, 8, 8).double().mean(0).cpu().numpy() / nk
    tot = t[:, :len(NAMES)].sum(1)
    print(f"mode {os.environ.get('ST_GU512_MODE', '0')}, {M} rows: {tot.mean():.0f} cycles per K-step per wave (MFMA work of a SIMD's two waves: 1280); "
          f"{us:.1f} us per traced launch = {tot.mean() * nk / us / 1e3:.2f} GHz if the loop is the launch")
    print("    " + " " * 44 + "".join(f"   wave{w}" for w in range(8)))
    for i, nm in enumerate(NAMES):
        print(f"    {nm:44s}" + "".join(f"{t[w, i]:8.0f}" for w in range(8)))


main()
