#!/bin/bash
# one-pass quantiser for both gradient GEMMs: fp8 tests, fp8 learning test, bench fp8 forward + dgrad + wgrad (default workload)
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_fp8.py -x -q 2>&1 | tail -3
timeout 600 python -m pytest tests/test_gpu_e2e.py -q -s -k "fp8_mode" 2>&1 | grep "share of sampled\|passed\|failed"
timeout 1200 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --dtype fp8 --fp8-dgrad --fp8-wgrad > gpurun_out/r04/bench_fp8all_c.json 2> gpurun_out/r04/bench_fp8all_c.err
python3 -c "
import json
d = json.loads(open('gpurun_out/r04/bench_fp8all_c.json').read().strip().splitlines()[-1])
print(d['value'], d['timing_s'], d.get('peak_mem_gb'))"
