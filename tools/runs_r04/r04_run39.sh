#!/bin/bash
# one H2D copy per image and step (device cache of the staged pixel tensors): tests that stage images, then the default bench line
mkdir -p gpurun_out/r04
timeout 1800 python -m pytest tests/test_gpu_rollout.py tests/test_gpu_e2e.py tests/test_gpu_model.py tests/test_gpu_token_budget.py tests/test_gpu_critic.py -x -q 2>&1 | tail -3
timeout 900 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r04/bench_i.json 2> gpurun_out/r04/bench_i.err
python3 -c "
import json
d = json.loads(open('gpurun_out/r04/bench_i.json').read().strip().splitlines()[-1])
print(d['value'], d['timing_s'], d['roofline']['frac'])"
