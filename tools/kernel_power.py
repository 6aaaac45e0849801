"""Shader clock and socket power under ONE kernel class at a time (bench.py's ChipTelemetry: st_clock_probe on a side stream + the GPU's
hwmon files), each class looped back to back for a few seconds:  python tools/kernel_power.py [seconds per class]

What it is for: the step runs at 1.34 kW of a 1.4-kW cap (profiles/r06_notes.md §0); this says which kernels hold the chip at the cap and at
what clock, i.e. what a cycle is worth in each of them."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from spatialthinker_amd import ops  # noqa: E402
from spatialthinker_amd.lib import lib  # noqa: E402

SECONDS = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
dev = torch.device("cuda:0")
torch.manual_seed(0)
BF = torch.bfloat16


def rnd(*shape, scale=0.05):
    return (torch.randn(*shape, device=dev) * scale).to(BF)


def training_gemm():
    a, w = rnd(20864, 3584, scale=0.5), rnd(37888, 3584)
    out = torch.empty(20864, 37888, dtype=BF, device=dev)
    flops = 2.0 * 20864 * 37888 * 3584
    return (lambda: ops.gemm_nt(a, w, out=out)), flops, "training GEMM (st_gemm_nt 20864 x 37888 x 3584)"


def decode_gate_up(rows):
    a = rnd(rows, 3584, scale=0.5)
    ws = [rnd(2 * 18944, 3584) for _ in range(6)]
    out = torch.empty(rows, 18944, dtype=BF, device=dev)
    i = [0]

    def f():
        i[0] += 1
        ops.gemm_swiglu_decode(a, ws[i[0] % 6], out=out)
    return f, 2.0 * rows * 2 * 18944 * 3584, f"decode gate/up + SwiGLU at {rows} rows (six rotating weight copies)"


def swiglu_backward():
    T, I = 8192, 18944
    gu, dy = rnd(T, 2 * I, scale=0.5), rnd(T, I, scale=0.5)
    d = torch.empty_like(gu)
    return (lambda: ops.swiglu_bwd(gu, dy, out=d)), 0.0, f"swiglu_bwd ({T} x {I}: HBM-bound elementwise, {5 * T * I * 2 / 1e9:.2f} GB per launch)"


def rmsnorm_forward():
    T, H = 65536, 3584
    x, w = rnd(T, H, scale=0.5), rnd(H, scale=1.0)
    y = torch.empty_like(x)
    return (lambda: ops.rmsnorm_fwd(x, w, 1e-6, want_rstd=False, out=y)), 0.0, f"rmsnorm forward ({T} x {H})"


def attention_forward():
    S, nq, nkv, D = 4096, 28, 4, 128
    q, k, v = rnd(S * 4, nq * D, scale=0.5), rnd(S * 4, nkv * D, scale=0.5), rnd(S * 4, nkv * D, scale=0.5)
    cu = torch.tensor([0, S, 2 * S, 3 * S, 4 * S], dtype=torch.int32, device=dev)
    flops = 4.0 * 4 * nq * D * S * S / 2
    o = torch.empty(S * 4, nq * D, dtype=BF, device=dev)
    return (lambda: ops.attn_fwd(q, k, v, cu, S, nq, nkv, D, D ** -0.5, True, out=o)), flops, f"attention forward, causal, 4 x {S} tokens, D = 128"


def attention_backward():
    S, nq, nkv, D = 4096, 28, 4, 128
    q, k, v = rnd(S * 4, nq * D, scale=0.5), rnd(S * 4, nkv * D, scale=0.5), rnd(S * 4, nkv * D, scale=0.5)
    cu = torch.tensor([0, S, 2 * S, 3 * S, 4 * S], dtype=torch.int32, device=dev)
    o, lse = ops.attn_fwd(q, k, v, cu, S, nq, nkv, D, D ** -0.5, True)[:2]
    do = rnd(S * 4, nq * D, scale=0.5)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    flops = 10.0 * 4 * nq * D * S * S / 2
    return (lambda: ops.attn_bwd(q, k, v, o, do, lse, cu, S, nq, nkv, D, D ** -0.5, True, dq, dk, dv)), flops, f"attention backward (three kernels), causal, 4 x {S} tokens"


def decode_narrow(rows):
    a, w = rnd(rows, 18944, scale=0.5), [rnd(3584, 18944) for _ in range(6)]
    out = torch.empty(rows, 3584, dtype=BF, device=dev)
    i = [0]

    def f():
        i[0] += 1
        ops.gemm_nt(a, w[i[0] % 6], out=out, decode=True)
    return f, 2.0 * rows * 3584 * 18944, f"decode down projection at {rows} rows (split-K tiles + finish, six rotating weight copies)"


CASES = [("idle", None), ("gemm", training_gemm), ("gu512", lambda: decode_gate_up(512)), ("gu256", lambda: decode_gate_up(256)),
         ("swiglu_bwd", swiglu_backward), ("rmsnorm", rmsnorm_forward), ("attention", attention_forward), ("attention_bwd", attention_backward), ("down512", lambda: decode_narrow(512)),
         ("down64", lambda: decode_narrow(64))]


def main():
    tel = bench.ChipTelemetry(period=0.1)
    tel.start()
    rows = {}
    for name, make in CASES:
        if make is None:
            tel.phase = name
            time.sleep(SECONDS)
            rows[name] = {"what": "nothing queued"}
            continue
        try:
            fn, flops, what = make()
        except Exception as e:                                # an op whose python signature differs on this tree: skip the class, keep the rest
            rows[name] = {"what": f"skipped ({type(e).__name__}: {e})"}
            continue
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        tel.phase = "ramp"
        t_end = time.perf_counter() + 1.0                     # a second of the same load before the labelled samples: the clock settles
        while time.perf_counter() < t_end:
            for _ in range(8):
                fn()
            torch.cuda.synchronize()
        tel.phase = name
        n, e0, e1 = 0, torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        t_end = time.perf_counter() + SECONDS
        while time.perf_counter() < t_end:
            for _ in range(8):
                fn()
            n += 8
            torch.cuda.synchronize()
        e1.record(); torch.cuda.synchronize()
        tel.phase = "between"
        us = e0.elapsed_time(e1) * 1e3 / n
        rows[name] = {"what": what, "us_per_launch": us, "tflops": flops / us * 1e-6 if flops else None}
        time.sleep(0.5)
    tel.stop()
    s = tel.summary(gemm_phases=("gemm",))
    print(f"power cap {s.get('power_cap_w')} W, pci {s.get('pci')}")
    print(f"{'class':12s} {'clock MHz (mean / min)':>24s} {'socket W (mean / max)':>22s} {'us per launch':>14s} {'TF/s':>7s}  {'pJ per flop':>12s}  what")
    for name, _ in CASES:
        st, r = (s["by_phase"].get(name) or {}), rows.get(name, {})
        w = st.get("socket_power_w_mean")
        tf = r.get("tflops")
        print(f"{name:12s} {st.get('clock_mhz_mean', float('nan')):12.0f} / {st.get('clock_mhz_min', float('nan')):<9.0f} "
              f"{(w if w is not None else float('nan')):10.0f} / {st.get('socket_power_w_max', float('nan')):<9.0f} "
              f"{r.get('us_per_launch', float('nan')):14.1f} {(tf if tf else float('nan')):7.0f}  "
              f"{(w / tf if (w and tf) else float('nan')):12.2f}  {r.get('what', '')}")
    if len(sys.argv) > 2:
        json.dump({"rows": rows, "telemetry": s}, open(sys.argv[2], "w"), indent=1)


main()
