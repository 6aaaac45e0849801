"""dX / dW GEMMs of a 7B layer: contraction-major kernels (st_gemm_nn / st_gemm_tn) vs transposes + the NT kernel, interleaved:
python tools/gemm_layout_ab.py [T]"""
import os, sys, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatialthinker_amd import ops
T = int(sys.argv[1]) if len(sys.argv) > 1 else 10496
H, QKV, I2, I = 3584, 4608, 37888, 18944
def timeit(fn, iters=5):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
tot = [0.0, 0.0, 0.0]
for name, Nout, Kin in [("qkv", QKV, H), ("o", H, H), ("gu", I2, H), ("down", H, I)]:
    dy = torch.randn(T, Nout, device="cuda").bfloat16(); x = torch.randn(T, Kin, device="cuda").bfloat16()
    w = (torch.randn(Nout, Kin, device="cuda") * 0.02).bfloat16(); wT = ops.transpose(w)
    gw = torch.zeros(Nout, Kin, device="cuda"); dx = torch.empty(T, Kin, device="cuda", dtype=torch.bfloat16)
    runs = {
        "dx  new": lambda: ops.gemm_nn(dy, w, out=dx),
        "dx  old": lambda: ops.gemm_nt(dy, wT, out=dx),                       # transposed weight copy kept up to date elsewhere
        "dw  new": lambda: ops.gemm_tn(dy, x, gw, accumulate=True),
        "dw  old": lambda: ops.gemm_nt(ops.transpose(dy), ops.transpose(x), out_f32=gw, accumulate=True),
    }
    for f in runs.values(): f()
    ts = {k: [] for k in runs}
    for _ in range(5):
        for k, f in runs.items(): ts[k].append(timeit(f))
    med = {k: statistics.median(v) for k, v in ts.items()}
    fl = 2.0 * T * Nout * Kin / 1e6
    print(f"{name:5s}: dX nn {med['dx  new']:7.1f} us ({fl / med['dx  new']:5.0f} TF) vs nt+wT {med['dx  old']:7.1f} us ({fl / med['dx  old']:5.0f} TF) | "
          f"dW tn {med['dw  new']:7.1f} us ({fl / med['dw  new']:5.0f} TF) vs transposes+nt {med['dw  old']:7.1f} us ({fl / med['dw  old']:5.0f} TF)", flush=True)
    tot[0] += med["dx  new"] + med["dw  new"]; tot[1] += med["dx  old"] + med["dw  old"]
    del dy, x, w, wT, gw, dx
print(f"layer backward GEMMs: contraction-major {tot[0] / 1e3:.2f} ms vs transposes + NT {tot[1] / 1e3:.2f} ms ({100 * (tot[1] / tot[0] - 1):+.1f} %)")
