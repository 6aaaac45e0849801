#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python bench.py --steps 16 --warmup 1 --no-cpu-baseline > gpurun_out/r03_bench_long.json 2> gpurun_out/r03_bench_long.err; python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03_bench_long.json").read().strip().splitlines()[-1])
print(d["value"], d["steps"], {k: round(v, 2) for k, v in d["timing_s"].items()}, "peak", d["peak_mem_gb"], d["peak_reserved_gb"], d["reserved_gb_after_each_step"])
PY
