#!/bin/bash
# fp8: transposing quantiser on 128x128 tiles; tests, then kernel stats + bench line of fp8 forward + dgrad + wgrad (default workload and cfg #5's shape)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_fp8.py -x -q 2>&1 | tail -3
rm -rf /tmp/prof_fp8
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_fp8 -o x -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --dtype fp8 --fp8-dgrad --fp8-wgrad > gpurun_out/r04/fp8_bench_prof.json 2> /tmp/prof_fp8.err
cp /tmp/prof_fp8/x_kernel_stats.csv gpurun_out/r04/fp8_bench_kernel_stats.csv
for shape in "" "--rollouts 16 --prompts-per-gpu 32 --image 896x896"; do
  tag=$(echo "x$shape" | tr -d ' -' | cut -c1-12)
  timeout 1200 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline $shape --dtype fp8 --fp8-dgrad --fp8-wgrad > gpurun_out/r04/bench_fp8all_$tag.json 2> gpurun_out/r04/bench_fp8all_$tag.err
  python3 -c "
import json
d = json.loads(open('gpurun_out/r04/bench_fp8all_$tag.json').read().strip().splitlines()[-1])
print('$shape', d['value'], d['timing_s'], d.get('peak_mem_gb'))
for c in d.get('roofline_classes', []):
    if 'mx4' in c['kernel']: print('   ', c['kernel'][:40], c['achieved'], c['frac'])"
done
