#!/bin/bash
# fp8 weight gradients: transposing quantiser, fp32-accumulating fp8 tile, layer test, learning test, bench fp8 + dgrad (+ wgrad) on cfg #5's shape
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_fp8.py -x -q -s 2>&1 | grep -v "^$" | grep -v "MX-fp8 GEMM" | tail -12
timeout 600 python -m pytest tests/test_gpu_e2e.py -q -s -k "fp8_mode" 2>&1 | grep "share of sampled\|passed\|failed"
for mode in "--dtype fp8 --fp8-dgrad" "--dtype fp8 --fp8-dgrad --fp8-wgrad"; do
  tag=$(echo $mode | tr -d ' -')
  timeout 1200 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --rollouts 16 --prompts-per-gpu 32 --image 896x896 $mode > gpurun_out/r04/bench_cfg5b_$tag.json 2> gpurun_out/r04/bench_cfg5b_$tag.err
  python3 -c "
import json
d = json.loads(open('gpurun_out/r04/bench_cfg5b_$tag.json').read().strip().splitlines()[-1])
print('$mode', d['value'], d['timing_s'], d.get('peak_mem_gb'))"
done
