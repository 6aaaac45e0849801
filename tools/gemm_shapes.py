"""All twelve GEMMs of one 7B LM layer (forward, dX, dW) through the entry points the engine uses (st_gemm_nt with its fused
epilogues, st_gemm_swiglu, st_gemm_tn) at the packed-token count of a fused update pass: production tile (variant 40) vs the 8-wave
tile (variant 23) vs hipBLASLt (torch.matmul, plain product).  TF/s count the matmul flops only.
    python tools/gemm_shapes.py [T ...]      default T = 21504"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatialthinker_amd import ops  # noqa: E402


def bench(fn, iters=6):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


H, QKV, I = 3584, 4608, 18944
ops._gemm_workspace(torch.device("cuda"))
rb = lambda *s: (torch.randn(*s, device="cuda") * 0.1).bfloat16()
for T in [int(a) for a in sys.argv[1:]] or [21504]:
    x, bias = rb(T, H), rb(QKV)
    res = rb(T, H)
    w = {"qkv": rb(QKV, H), "o": rb(H, H), "gu": rb(2 * I, H), "down": rb(H, I)}
    wT = {k: ops.transpose(v) for k, v in w.items()}
    m = rb(T, I)
    dy = {"qkv": rb(T, QKV), "o": rb(T, H), "gu": rb(T, 2 * I), "down": rb(T, H)}
    gw = {k: torch.zeros(v.shape, dtype=torch.float32, device="cuda") for k, v in w.items()}
    cases = [("qkv+bias", 2.0 * T * QKV * H, lambda: ops.gemm_nt(x, w["qkv"], bias=bias), lambda: torch.matmul(x, w["qkv"].t())),
             ("o+res", 2.0 * T * H * H, lambda: ops.gemm_nt(x, w["o"], residual=res), lambda: torch.matmul(x, w["o"].t())),
             ("gateup+swiglu", 2.0 * T * 2 * I * H, lambda: ops.gemm_swiglu(x, w["gu"], want_gu=True), lambda: torch.matmul(x, w["gu"].t())),
             ("down+res", 2.0 * T * H * I, lambda: ops.gemm_nt(m, w["down"], residual=res), lambda: torch.matmul(m, w["down"].t())),
             ("dx_qkv", 2.0 * T * H * QKV, lambda: ops.gemm_nt(dy["qkv"], wT["qkv"]), lambda: torch.matmul(dy["qkv"], w["qkv"])),
             ("dx_o", 2.0 * T * H * H, lambda: ops.gemm_nt(dy["o"], wT["o"]), lambda: torch.matmul(dy["o"], w["o"])),
             ("dx_gu", 2.0 * T * H * 2 * I, lambda: ops.gemm_nt(dy["gu"], wT["gu"]), lambda: torch.matmul(dy["gu"], w["gu"])),
             ("dx_down", 2.0 * T * I * H, lambda: ops.gemm_nt(dy["down"], wT["down"]), lambda: torch.matmul(dy["down"], w["down"])),
             ("dw_qkv", 2.0 * T * QKV * H, lambda: ops.gemm_tn(dy["qkv"], x, gw["qkv"], accumulate=True), lambda: torch.matmul(dy["qkv"].t(), x)),
             ("dw_o", 2.0 * T * H * H, lambda: ops.gemm_tn(dy["o"], x, gw["o"], accumulate=True), lambda: torch.matmul(dy["o"].t(), x)),
             ("dw_gu", 2.0 * T * 2 * I * H, lambda: ops.gemm_tn(dy["gu"], x, gw["gu"], accumulate=True), lambda: torch.matmul(dy["gu"].t(), x)),
             ("dw_down", 2.0 * T * H * I, lambda: ops.gemm_tn(dy["down"], m, gw["down"], accumulate=True), lambda: torch.matmul(dy["down"].t(), m))]
    tot = {"v40": 0.0, "v23": 0.0, "lib": 0.0}
    for name, fl, ours, lib_ in cases:
        ops.gemm_select(40); t40 = bench(ours)
        ops.gemm_select(23); t23 = bench(ours)
        ops.gemm_select(40)
        tl = bench(lib_)
        tot["v40"] += t40; tot["v23"] += t23; tot["lib"] += tl
        print(f"T={T} {name:14s}: v40 {fl / t40 / 1e12:6.0f}  v23 {fl / t23 / 1e12:6.0f}  hipblaslt {fl / tl / 1e12:6.0f} TF/s   ({t40 * 1e6:7.0f} us)", flush=True)
    # dX through the weight AS STORED (st_gemm_nn, B contraction-major) against the same product through a transposed copy (NT)
    for nm, K_, N_ in (("qkv", QKV, H), ("o", H, H), ("gu", 2 * I, H), ("down", H, I)):
        fl = 2.0 * T * K_ * N_
        t_nt, t_nn = bench(lambda: ops.gemm_nt(dy[nm], wT[nm])), bench(lambda: ops.gemm_nn(dy[nm], w[nm]))
        print(f"T={T} dx_{nm:5s} NN (weights as stored) {fl / t_nn / 1e12:6.0f} TF/s   vs NT through the transposed copy {fl / t_nt / 1e12:6.0f} TF/s", flush=True)
    fl_all = sum(c[1] for c in cases)
    print(f"T={T} layer total: v40 {tot['v40'] * 1e3:.2f} ms ({fl_all / tot['v40'] / 1e12:.0f} TF/s), v23 {tot['v23'] * 1e3:.2f} ms, hipblaslt (plain products) {tot['lib'] * 1e3:.2f} ms")
