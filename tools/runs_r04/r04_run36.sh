#!/bin/bash
# BASELINE config #2 (3B, 448x448) on the final sources: bf16 and fp8 (forward + dgrad + wgrad)
mkdir -p gpurun_out/r04
for mode in "--dtype bf16" "--dtype fp8 --fp8-dgrad --fp8-wgrad"; do
  tag=$(echo $mode | tr -d ' -')
  timeout 900 python3 bench.py --model 3b --image 448x448 --steps 2 --warmup 1 --no-cpu-baseline $mode > gpurun_out/r04/bench_cfg2_$tag.json 2> gpurun_out/r04/bench_cfg2_$tag.err
  python3 -c "
import json
d = json.loads(open('gpurun_out/r04/bench_cfg2_$tag.json').read().strip().splitlines()[-1])
print('$mode', d['value'], d['timing_s'], d.get('peak_mem_gb'))"
done
