#!/bin/bash
# swiglu backward that recomputes m in the same pass: kernel / model / trajectory tests, bench line
mkdir -p gpurun_out/r04
timeout 1800 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_trajectory.py tests/test_gpu_train_options.py tests/test_gpu_fp8.py -x -q 2>&1 | tail -4
timeout 900 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r04/bench_g.json 2> gpurun_out/r04/bench_g.err
python3 -c "
import json
d = json.loads(open('gpurun_out/r04/bench_g.json').read().strip().splitlines()[-1])
print(d['value'], d['timing_s'], d['roofline']['frac'])"
