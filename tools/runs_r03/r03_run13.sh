#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_byname_api.py tests/test_gpu_token_budget.py tests/test_gpu_e2e.py tests/test_gpu_dp.py -m gpu -q > gpurun_out/r03_gputests_13.log 2>&1; echo "pytest rc=$?"
tail -6 gpurun_out/r03_gputests_13.log
python bench.py --steps 2 --warmup 1 > gpurun_out/r03_bench_e.json 2> gpurun_out/r03_bench_e.err; tail -2 gpurun_out/r03_bench_e.err; python -c "
import json;d=json.load(open('gpurun_out/r03_bench_e.json'));print(d['value'], d['timing_s'], d['roofline']['frac'], d['roofline'].get('algorithmic_bytes_per_launch'), d['roofline_decode']['hbm'], d['roofline_decode']['mfma'], d['cpu_baseline']['value'], d['cpu_baseline']['cores'], d['cpu_baseline']['measured_s'])"
python bench.py --through-api --steps 2 --warmup 1 > gpurun_out/r03_bench_api.json 2> gpurun_out/r03_bench_api.err; tail -3 gpurun_out/r03_bench_api.err; cat gpurun_out/r03_bench_api.json | head -c 1500
