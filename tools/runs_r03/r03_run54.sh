#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for u in 1 0; do
ST_SWIGLU_UNFUSED_GRAD=$u timeout 900 python bench.py --no-cpu-baseline > gpurun_out/r03_bench_sw$u.json 2> gpurun_out/r03_bench_sw$u.err; python - <<PY
import json
d = json.loads(open("gpurun_out/r03_bench_sw$u.json").read().strip().splitlines()[-1])
print("unfused=$u", round(d["value"], 3), {k: round(v, 2) for k, v in d["timing_s"].items()}, d["peak_mem_gb"], round(d["roofline"]["frac"], 4), round(d["roofline"]["achieved"], 1))
PY
done
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_fullsize.py tests/test_gpu_token_budget.py -x -q -m gpu 2>&1 | tail -3
