#!/bin/bash
# fp8: fused producers (rmsnorm / swiglu / swiglu epilogue -> fp8), fp8 dgrad; tests, then bench fp8 and fp8 + dgrad, cfg #5 shape bf16 vs fp8
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_fp8.py -x -q -s 2>&1 | grep -v "^$" | tail -25
for mode in "--dtype fp8" "--dtype fp8 --fp8-dgrad"; do
  tag=$(echo $mode | tr -d ' -')
  timeout 900 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline $mode > gpurun_out/r04/bench_e_$tag.json 2> gpurun_out/r04/bench_e_$tag.err
  python3 -c "
import json
d = json.loads(open('gpurun_out/r04/bench_e_$tag.json').read().strip().splitlines()[-1])
print('$mode', d['value'], d['timing_s'], d.get('peak_mem_gb'))"
done
