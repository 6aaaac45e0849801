#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -q --durations=10 > gpurun_out/r03_gputests_2.log 2>&1; echo "pytest rc=$?"
tail -40 gpurun_out/r03_gputests_2.log
python tools/gemm_sustained.py 4 > gpurun_out/r03_gemm_sustained.log 2>&1; cat gpurun_out/r03_gemm_sustained.log
python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r03_bench_b.json 2> gpurun_out/r03_bench_b.err; tail -c 600 gpurun_out/r03_bench_b.err; python -c "
import json;d=json.load(open('gpurun_out/r03_bench_b.json'));print(d['value'], d['timing_s'], d['peak_mem_gb'], d['peak_reserved_gb'], d['reserved_gb_after_each_step'], d['passes_per_step'], d['roofline_decode']['ms_per_iteration'], d['roofline']['frac'])"
ST_DECODE_NT=0 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r03_bench_b_nont.json 2> gpurun_out/r03_bench_b_nont.err; python -c "
import json;d=json.load(open('gpurun_out/r03_bench_b_nont.json'));print('no-nt', d['value'], d['timing_s'], d['roofline_decode']['ms_per_iteration'])"
