#!/bin/bash
# full GPU suite after the fp8 work + the fp8 learning test; default bench line (bf16 headline)
mkdir -p gpurun_out/r04
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -8
timeout 600 python -m pytest tests/test_gpu_e2e.py -q -s -k "learns" 2>&1 | grep "share of sampled"
timeout 900 python3 bench.py > gpurun_out/r04/bench_f.json 2> gpurun_out/r04/bench_f.err
python3 -c "
import json
d = json.loads(open('gpurun_out/r04/bench_f.json').read().strip().splitlines()[-1])
print(d['value'], d['timing_s'], d['roofline']['frac'], d['cpu_baseline']['value'], d['roofline_decode']['ms_per_iteration'])"
