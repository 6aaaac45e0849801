#!/bin/bash
# the driver's bench command again: cpu_baseline with two cores of the quota left free + throttle counters per leg
mkdir -p gpurun_out/r04
timeout 1500 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04/bench_final_20e.json 2> gpurun_out/r04/bench_final_20e.err
python3 -c "
import json
d = json.loads(open('gpurun_out/r04/bench_final_20e.json').read().strip().splitlines()[-1])
c = d['cpu_baseline']
print('20 steps:', d['value'], d['timing_s'], 'cpu', c['value'], c['value_excl_generation_and_adamw'], c['cores'])
print(json.dumps(c['measured_s']))
print(json.dumps(c['host']))"
