#!/bin/bash
# fp8: SwiGLU epilogue of the 4-wave tile (bit-identity test), layer test, bench --dtype fp8
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_fp8.py -x -q 2>&1 | tail -8
timeout 900 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --dtype fp8 > gpurun_out/r04/bench_d_fp8.json 2> gpurun_out/r04/bench_d_fp8.err
python3 -c "
import json
d = json.loads(open('gpurun_out/r04/bench_d_fp8.json').read().strip().splitlines()[-1])
print('fp8', d['value'], d['timing_s'])"
