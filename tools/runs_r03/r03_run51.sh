#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r03_gputests_51.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r03_gputests_51.log
timeout 900 python bench.py > gpurun_out/r03_bench_j.json 2> gpurun_out/r03_bench_j.err; python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03_bench_j.json").read().strip().splitlines()[-1])
print(d["value"], {k: round(v, 2) for k, v in d["timing_s"].items()}, d["peak_mem_gb"], d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline_decode"]["frac"], d["roofline_decode"]["traffic"], d["cpu_baseline"]["value"])
PY
