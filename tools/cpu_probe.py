import os, time, torch
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "torch threads default", torch.get_num_threads())
try:
    print("cgroup cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip())
except Exception as e:
    print("no cgroup v2 cpu.max", e)
a = torch.randn(1614, 3584); b = torch.randn(18944, 3584)
for nt in (torch.get_num_threads(), 16, 32, 64, 128):
    torch.set_num_threads(nt)
    a @ b.t()
    t0 = time.perf_counter(); a @ b.t(); t = time.perf_counter() - t0
    print(nt, "threads: matmul", t, "s ->", 2 * 1614 * 3584 * 18944 / t / 1e12, "TF")
