"""Host-side probe for the cpu_baseline legs (python tools/cpu_probe.py): what the box gives this process (cpu count, affinity, cgroup
quota and throttle counters) and how the two small-op legs (a batch-8 matvec chain like the CPU decode step, bf16 elementwise ops like the
CPU AdamW) behave with 4 / 8 / 16 threads — they moved 4-10x between boxes while the large matmuls stayed within 6 %."""
import os, time, torch

def cg(name):
    try:
        return open("/sys/fs/cgroup/" + name).read().strip().replace("\n", " ")
    except Exception as e:
        return f"n/a ({type(e).__name__})"
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "torch threads default", torch.get_num_threads(), "interop", torch.get_num_interop_threads())
print("cgroup cpu.max:", cg("cpu.max"), "| cpu.stat:", cg("cpu.stat"))
print("loadavg", open("/proc/loadavg").read().strip())
print("env", {k: v for k, v in os.environ.items() if k.startswith(("OMP_", "MKL_", "KMP_", "GOMP_"))})
H, I = 2048, 11008
ws = [torch.randn(2560, H), torch.randn(H, H), torch.randn(2 * I, H), torch.randn(H, I)]
x = torch.randn(8, H)
def chain():
    h = x
    for _ in range(4):
        q = h @ ws[0].t(); o = q[:, :H] @ ws[1].t(); g = o @ ws[2].t(); h = (g[:, :I] * g[:, I:]) @ ws[3].t()
    return h
pa = torch.randn(1 << 24).bfloat16(); ga = torch.randn(1 << 24).bfloat16(); ma = torch.zeros_like(pa)
def elem():
    ma.mul_(0.9).add_(ga, alpha=0.1); pa.add_(ma, alpha=-1e-6)
big_a = torch.randn(1614, 3584); big_b = torch.randn(4608, 3584)
for rnd in range(2):
    for nt in (4, 8, 16):
        torch.set_num_threads(nt)
        out = []
        for fn, n in ((chain, 5), (elem, 5), (lambda: big_a @ big_b.t(), 3)):
            fn()
            ts = []
            for _ in range(n):
                t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
            out.append(f"min {min(ts)*1e3:7.1f} med {sorted(ts)[len(ts)//2]*1e3:7.1f} ms")
        print(f"round {rnd} threads {nt:2d}: matvec chain {out[0]} | bf16 elementwise {out[1]} | matmul {out[2]}")
    print("   cpu.stat:", cg("cpu.stat"), "| loadavg", open("/proc/loadavg").read().strip())
