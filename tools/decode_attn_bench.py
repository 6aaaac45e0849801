"""Decode attention (one launch of st_attn_fwd_ranges per layer: prompt partials + generated partials) at the bench's shapes, split by item
kind.  B sequences = P prompts x n rollouts, prompt length Lp (shared K/V), generated context ctx per sequence.

    python tools/decode_attn_bench.py [P] [n] [Lp] [ctx]
"""
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("ST_LIB"):
    import spatialthinker_amd.lib as _lib
    _lib.LIB_PATH = os.path.abspath(os.environ["ST_LIB"])
from spatialthinker_amd import ops  # noqa: E402

I32, BF16, F32 = torch.int32, torch.bfloat16, torch.float32


def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        with torch.cuda.graph(g, stream=st):
            for _ in range(n):
                fn()
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (3 * n)


def main():
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 43
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    Lp = int(sys.argv[3]) if len(sys.argv) > 3 else 1152
    ctx = int(sys.argv[4]) if len(sys.argv) > 4 else 256
    nq, nkv, D, R, CK, CKG = 28, 4, 128, 2048, 256, 512
    g = nq // nkv
    B = P * n
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    width = nkv * D
    NL = 6                                                                 # rotate over several layers' worth of K/V: cold data
    kp = [torch.randn(P * Lp, width, device=dev).to(BF16) for _ in range(NL)]
    vp = [torch.randn(P * Lp, width, device=dev).to(BF16) for _ in range(NL)]
    kg = [torch.randn(B * R, width, device=dev).to(BF16) for _ in range(NL)]
    vg = [torch.randn(B * R, width, device=dev).to(BF16) for _ in range(NL)]
    qkv = torch.randn(B, (nq + 2 * nkv) * D, device=dev).to(BF16)
    C, Cg = -(-Lp // CK), -(-R // CKG)
    NP = C + Cg
    rows_all = B * g
    ar = torch.arange(B, device=dev, dtype=I32)
    pb = torch.arange(P, device=dev, dtype=I32) * Lp
    pe = pb + Lp
    first = torch.arange(P, device=dev, dtype=I32) * n
    kb1 = torch.cat([torch.minimum(pb + c * CK, pe) for c in range(C)]); ke1 = torch.cat([torch.minimum(pb + (c + 1) * CK, pe) for c in range(C)])
    qb1 = (first * g).repeat(C); qe1 = ((first + n) * g).repeat(C)
    ob1 = torch.cat([c * rows_all + first * g for c in range(C)]).to(I32)
    qb2 = (ar * g).repeat(Cg); qe2 = qb2 + g
    kbase = (ar * R).repeat(Cg)
    cidx = torch.arange(Cg, device=dev, dtype=I32).repeat_interleave(B)
    kb2 = kbase + cidx * CKG
    ke2 = torch.maximum(torch.minimum(kb2 + CKG, kbase + ctx), kb2)
    ob2 = ((C + cidx) * rows_all + (ar * g).repeat(Cg)).to(I32)
    parts = torch.empty(NP * rows_all, width, dtype=BF16, device=dev)
    lse = torch.empty(nkv, NP * rows_all, dtype=F32, device=dev)
    z1, z2 = torch.zeros_like(kb1), torch.zeros_like(kb2)
    scale = 1.0 / math.sqrt(D)
    cnt = [0]

    def both():
        cnt[0] += 1; L = cnt[0] % NL
        ops.attn_fwd_ranges(qkv, kg[L], vg[L], torch.cat([qb1, qb2]), torch.cat([qe1, qe2]), torch.cat([z1, kb2]), torch.cat([z1, ke2]), max(n * g, g), nkv, nkv,
                            D, scale, parts, lse, o_beg=torch.cat([ob1, ob2]), q_group=g, pre_beg=torch.cat([kb1, z2]), pre_end=torch.cat([ke1, z2]),
                            k_pre=kp[L], v_pre=vp[L])
    args_all = [torch.cat(x).contiguous() for x in ((qb1, qb2), (qe1, qe2), (z1, kb2), (z1, ke2), (ob1, ob2), (kb1, z2), (ke1, z2))]

    def run(qb, qe, kb, ke, ob, pb_, pe_, mq):
        def f():
            cnt[0] += 1; L = cnt[0] % NL
            ops.attn_fwd_ranges(qkv, kg[L], vg[L], qb, qe, kb, ke, mq, nkv, nkv, D, scale, parts, lse, o_beg=ob, q_group=g, pre_beg=pb_, pre_end=pe_,
                                k_pre=kp[L], v_pre=vp[L])
        return f
    t_all = timeit(run(*args_all, max(n * g, g)))
    t_prompt = timeit(run(qb1, qe1, z1, z1, ob1, kb1, ke1, n * g))
    t_own = timeit(run(qb2, qe2, kb2, ke2, ob2, z2, z2, g))
    by_p, by_o = P * Lp * width * 2 * 2, B * ctx * width * 2 * 2
    print(f"B = {P} x {n} = {B} rows, prompt {Lp}, generated context {ctx}: K/V bytes prompt {by_p / 1e6:.0f} MB + own {by_o / 1e6:.0f} MB")
    print(f"  one launch (as shipped): {t_all * 1e6:7.1f} us = {(by_p + by_o) / t_all / 1e12:5.2f} TB/s")
    print(f"  prompt partials only   : {t_prompt * 1e6:7.1f} us = {by_p / t_prompt / 1e12:5.2f} TB/s   ({qb1.numel()} items x {nkv} kv heads)")
    print(f"  generated partials only: {t_own * 1e6:7.1f} us = {by_o / t_own / 1e12:5.2f} TB/s   ({qb2.numel()} items x {nkv} kv heads, {int((ke2 > kb2).sum())} non-empty)")


if __name__ == "__main__":
    main()
