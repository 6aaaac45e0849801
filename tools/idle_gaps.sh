# where the GPU idles inside a GRPO step: rocprofv3 kernel trace of `bench.py --steps 1 --warmup 1`, gaps between consecutive kernels
#   bash tools/idle_gaps.sh <tag>      (from the repo root on the GPU box)
tag=${1:-r06}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -rf /tmp/idle_$tag
rocprofv3 --kernel-trace --output-format csv -d /tmp/idle_$tag -o x -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-fp8-leg --no-telemetry > /tmp/idle_$tag.json 2> /tmp/idle_$tag.err
tail -c 400 /tmp/idle_$tag.json
python3 - $tag <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
f = glob.glob(f'/tmp/idle_{tag}/**/x_kernel_trace.csv', recursive=True)[0]
ev = []
for r in csv.DictReader(open(f)):
    ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].replace('void ', '').replace('at::native::', '')[:90]))
ev.sort()
first = next(i for i, e in enumerate(ev) if 'sample_kernel' in e[2])
ev = ev[max(0, first - 2000):]
t0, t1 = ev[0][0], max(e[1] for e in ev)
busy_end = ev[0][1]; gaps = collections.defaultdict(lambda: [0, 0]); idle = 0
classes = [(20e3, '<20us'), (100e3, '20-100us'), (1e6, '0.1-1ms'), (10e6, '1-10ms'), (1e18, '>10ms')]
by_class = collections.defaultdict(lambda: [0, 0])
prev = ev[0][2]
for s, e, n in ev[1:]:
    if s > busy_end:
        g = s - busy_end
        idle += g
        for lim, name in classes:
            if g < lim:
                by_class[name][0] += 1; by_class[name][1] += g; break
        if g >= 20e3:
            k = (prev, n); gaps[k][0] += 1; gaps[k][1] += g
    if e > busy_end:
        busy_end = e; prev = n
print(f"window {(t1 - t0) / 1e9:.2f} s, idle {idle / 1e9:.3f} s = {100 * idle / (t1 - t0):.2f} %")
for name in [c[1] for c in classes]:
    print(f"  gaps {name:9s}: {by_class[name][0]:8d} x, {by_class[name][1] / 1e9:.3f} s")
big = []
be = ev[0][1]; pv = ev[0][2]
for i, (s_, e_, n_) in enumerate(ev[1:], 1):
    if s_ > be and s_ - be > 5e6:
        big.append((s_ - be, (be - t0) / 1e9, pv, [x[2][:44] for x in ev[i:i + 4]]))
    if e_ > be:
        be = e_; pv = n_
print("gaps > 5 ms in time order (offset from the window start):")
for g_, off, pv_, nx in big:
    print(f"  t={off:7.2f}s  {g_ / 1e6:7.1f} ms  after {pv_[:44]:44s} then {' | '.join(nx)}")
print("largest idle by (kernel before -> kernel after), gaps >= 20 us:")
for k, (c, g) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"  {g / 1e6:9.1f} ms in {c:5d} gaps: {k[0][:50]:50s} -> {k[1][:50]}")
PY
