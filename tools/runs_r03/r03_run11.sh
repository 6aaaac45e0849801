#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
ST_GEMM_VARIANT=40 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r03_bench_d_v40.json 2> gpurun_out/r03_bench_d_v40.err; tail -2 gpurun_out/r03_bench_d_v40.err; python -c "
import json;d=json.load(open('gpurun_out/r03_bench_d_v40.json'));print('v40', d['value'], d['timing_s'], d['peak_reserved_gb'], d['roofline_decode']['ms_per_iteration'], d['roofline']['frac'], d['roofline']['achieved'], d['actor_mfu'])"
python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r03_bench_d_v23.json 2> gpurun_out/r03_bench_d_v23.err; python -c "
import json;d=json.load(open('gpurun_out/r03_bench_d_v23.json'));print('v23', d['value'], d['timing_s'], d['peak_reserved_gb'], d['roofline_decode']['ms_per_iteration'], d['roofline']['frac'], d['roofline']['achieved'], d['actor_mfu'])"
