#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r03_gputests_32.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r03_gputests_32.log
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/r03_bench_g.json 2> gpurun_out/r03_bench_g.err; python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03_bench_g.json").read().strip().splitlines()[-1])
print(d["value"], d["timing_s"], d["peak_mem_gb"], d["roofline"]["frac"], d["roofline_decode"]["ms_per_iteration"], d["roofline_decode"]["frac"])
PY
