#!/bin/bash
# fp8 tests after the rmsnorm test fix; BASELINE config #5's per-GPU shape (G = 16, 896x896): bf16 vs fp8 + dgrad on one box
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_fp8.py -q -s 2>&1 | grep -v "^$" | tail -12
for mode in "--dtype bf16" "--dtype fp8 --fp8-dgrad"; do
  tag=$(echo $mode | tr -d ' -')
  timeout 1200 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --rollouts 16 --prompts-per-gpu 32 --image 896x896 $mode > gpurun_out/r04/bench_cfg5_$tag.json 2> gpurun_out/r04/bench_cfg5_$tag.err
  python3 -c "
import json
d = json.loads(open('gpurun_out/r04/bench_cfg5_$tag.json').read().strip().splitlines()[-1])
print('$mode', d['value'], d['timing_s'], d.get('peak_mem_gb'))
for c in d.get('roofline_classes', []):
    if 'mxfp8' in c['kernel']: print('   ', c['kernel'][:50], c['achieved'], c['frac'])"
done
