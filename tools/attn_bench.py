"""Attention kernel micro-benchmark + fp32 reference check on the training shape (packed causal GQA, D=128).

    python tools/attn_bench.py [n_seq] [seqlen]          (ST_ATTN_OLD=1 selects the previous forward kernel)
"""
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("ST_LIB"):                 # A/B a differently built library (tools/attn_ab.sh)
    import spatialthinker_amd.lib as _lib
    _lib.LIB_PATH = os.path.abspath(os.environ["ST_LIB"])
from spatialthinker_amd import ops  # noqa: E402


def ref_attn(q, k, v, cu, n_q, n_kv, D, scale, causal):
    outs = []
    g = n_q // n_kv
    for i in range(len(cu) - 1):
        a, b = cu[i], cu[i + 1]
        qq = q[a:b].float().view(b - a, n_q, D).transpose(0, 1)
        kk = k[a:b].float().view(b - a, n_kv, D).transpose(0, 1).repeat_interleave(g, 0)
        vv = v[a:b].float().view(b - a, n_kv, D).transpose(0, 1).repeat_interleave(g, 0)
        s = qq @ kk.transpose(1, 2) * scale
        if causal:
            m = torch.ones(b - a, b - a, dtype=torch.bool, device=q.device).tril()
            s = s.masked_fill(~m, float("-inf"))
        p = s.softmax(-1)
        outs.append((p @ vv).transpose(0, 1).reshape(b - a, n_q * D))
    return torch.cat(outs, 0)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def main():
    n_seq = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 1614
    n_q, n_kv, D = 28, 4, 128
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    lens = [S - 37 * i for i in range(n_seq)]
    cu_l = [0]
    for L in lens:
        cu_l.append(cu_l[-1] + L)
    T = cu_l[-1]
    cu = torch.tensor(cu_l, dtype=torch.int32, device=dev)
    qkv = torch.randn(T, (n_q + 2 * n_kv) * D, device=dev).to(torch.bfloat16)
    q, k, v = qkv[:, :n_q * D], qkv[:, n_q * D:(n_q + n_kv) * D], qkv[:, (n_q + n_kv) * D:]
    scale = 1.0 / math.sqrt(D)
    o, lse = ops.attn_fwd(q, k, v, cu, max(lens), n_q, n_kv, D, scale, True)
    ref = ref_attn(q, k, v, cu_l, n_q, n_kv, D, scale, True)
    err = (o.float() - ref).abs().max().item()
    print(f"fwd max err vs fp32 reference: {err:.3e} (ref max {ref.abs().max().item():.2f})")
    o2, _ = ops.attn_fwd(q, k, v, cu, max(lens), n_q, n_kv, D, scale, False)
    ref2 = ref_attn(q, k, v, cu_l, n_q, n_kv, D, scale, False)
    print(f"fwd (bidirectional) max err: {(o2.float() - ref2).abs().max().item():.3e}")

    flop_f = sum(4.0 * L * L * D * n_q / 2 for L in lens)
    t = timeit(lambda: ops.attn_fwd(q, k, v, cu, max(lens), n_q, n_kv, D, scale, True))
    print(f"fwd causal: {t * 1e6:8.1f} us  {flop_f / t / 1e12:7.1f} TF")
    t = timeit(lambda: ops.attn_fwd(q, k, v, cu, max(lens), n_q, n_kv, D, scale, False))
    print(f"fwd full  : {t * 1e6:8.1f} us  {2 * flop_f / t / 1e12:7.1f} TF")

    do = torch.randn(T, n_q * D, device=dev).to(torch.bfloat16)
    dq = torch.empty(T, n_q * D, dtype=torch.bfloat16, device=dev)
    dk = torch.empty(T, n_kv * D, dtype=torch.bfloat16, device=dev)
    dv = torch.empty(T, n_kv * D, dtype=torch.bfloat16, device=dev)
    ops.attn_bwd(q, k, v, o, do, lse, cu, max(lens), n_q, n_kv, D, scale, True, dq, dk, dv)
    # autograd reference
    qf = q.float().detach().requires_grad_(True)
    kf = k.float().detach().requires_grad_(True)
    vf = v.float().detach().requires_grad_(True)
    r = ref_attn(qf, kf, vf, cu_l, n_q, n_kv, D, scale, True)
    r.backward(do.float())
    for name, a, b in (("dq", dq, qf.grad), ("dk", dk, kf.grad), ("dv", dv, vf.grad)):
        print(f"{name} max err {(a.float() - b).abs().max().item():.3e} (ref max {b.abs().max().item():.2f})")
    t = timeit(lambda: ops.attn_bwd(q, k, v, o, do, lse, cu, max(lens), n_q, n_kv, D, scale, True, dq, dk, dv))
    print(f"bwd causal: {t * 1e6:8.1f} us  {2.5 * flop_f / t / 1e12:7.1f} TF (5-matmul flop count)")


if __name__ == "__main__":
    main()
