#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python bench.py --master-fp32 --no-cpu-baseline > gpurun_out/r03_bench_master.json 2> gpurun_out/r03_bench_master.err; python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r03_bench_master.json").read().strip().splitlines()[-1])
    print(d["value"], {k: round(v, 2) for k, v in d["timing_s"].items()}, d["peak_mem_gb"], d["peak_reserved_gb"], d["reserved_gb_after_each_step"])
except Exception as e:
    print("failed", e); print(open("gpurun_out/r03_bench_master.err").read()[-1500:])
PY
