"""Does the K/V cache LAYOUT bound the decode attention?  The same items (512 samples x 4 KV heads, 7 query rows, L keys each) read from
(a) the rollout's layout [sample*R + key][4 heads x 128] (a head's keys are 256-byte pieces at a 1-KiB pitch) and (b) a head-major layout
[head][sample*R + key][128] (a head's keys are one contiguous stream), through st_attn_decode_rows.      python tools/decode_attn_layout_probe.py [L]"""
import math, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatialthinker_amd import ops

L = int(sys.argv[1]) if len(sys.argv) > 1 else 256
B, R, nkv, g, D = 512, 1024, 4, 7, 128
dev = "cuda"
torch.manual_seed(0)
q = (torch.randn(B, nkv * g * D, device=dev) * 0.5).bfloat16()
ka = torch.randn(B * R, nkv * D, device=dev).bfloat16(); va = torch.randn(B * R, nkv * D, device=dev).bfloat16()
kb = ka.view(B * R, nkv, D).permute(1, 0, 2).contiguous(); vb = va.view(B * R, nkv, D).permute(1, 0, 2).contiguous()      # [head][row][128]
ar = torch.arange(B, device=dev, dtype=torch.int32)
qb, qe = (ar * g).contiguous(), (ar * g + g).contiguous()
kbeg, kend = (ar * R).contiguous(), (ar * R + L).contiguous()
scale = 1.0 / math.sqrt(D)


def run_a(slots):
    out = torch.empty(B * g, nkv * D, dtype=torch.bfloat16, device=dev); lse = torch.empty(nkv, B * g, dtype=torch.float32, device=dev)
    ops.attn_decode_rows(q, ka, va, qb, qe, kbeg, kend, g, nkv, D, scale, out, lse, q_group=g, slots=slots)
    return out


def run_b(slots):
    outs = []
    for h in range(nkv):                                   # one launch per head on its contiguous slab: q columns of head h, heads = 1
        qh = q.view(B, nkv, g * D)[:, h].contiguous()
        out = torch.empty(B * g, D, dtype=torch.bfloat16, device=dev); lse = torch.empty(1, B * g, dtype=torch.float32, device=dev)
        ops.attn_decode_rows(qh, kb[h], vb[h], qb, qe, kbeg, kend, g, 1, D, scale, out, lse, q_group=g, slots=slots)
        outs.append(out)
    return torch.stack(outs, 1).reshape(B * g, nkv * D)


def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


for slots in (2, 13):
    a, b = run_a(slots), run_b(slots)
    err = float((a.float() - b.float()).abs().max())
    qhs = [q.view(B, nkv, g * D)[:, h].contiguous() for h in range(nkv)]
    ta = timeit(lambda: run_a(slots))
    tb = timeit(lambda: run_b(slots))
    gb = B * nkv * L * D * 2 * 2 / 1e9
    print(f"L={L} slots={slots}: interleaved heads {ta * 1e6:.1f} us = {gb / ta / 1e3:.2f} TB/s | head-major (4 launches incl. q slicing) {tb * 1e6:.1f} us = {gb / tb / 1e3:.2f} TB/s | max diff {err:.3g}")
