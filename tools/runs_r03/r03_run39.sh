#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r03_gputests_39.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r03_gputests_39.log
timeout 900 python bench.py > gpurun_out/r03_bench_h.json 2> gpurun_out/r03_bench_h.err; python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03_bench_h.json").read().strip().splitlines()[-1])
print(d["value"], d["timing_s"], d["peak_mem_gb"], d["roofline"], d["roofline_decode"]["ms_per_iteration"], d["roofline_decode"]["frac"], d["cpu_baseline"])
PY
timeout 600 python bench.py --model 3b --image 448x448 --no-cpu-baseline > gpurun_out/r03_bench_cfg2.json 2> gpurun_out/r03_bench_cfg2.err; tail -c 300 gpurun_out/r03_bench_cfg2.json | head -c 300; echo
timeout 900 python bench.py --dtype fp8 --rollouts 16 --prompts-per-gpu 32 --image 896x896 --no-cpu-baseline > gpurun_out/r03_bench_cfg5.json 2> gpurun_out/r03_bench_cfg5.err; python - <<'PY'
import json
for f in ("cfg2", "cfg5"):
    try:
        d = json.loads(open(f"gpurun_out/r03_bench_{f}.json").read().strip().splitlines()[-1]); print(f, d["value"], d["timing_s"], d["peak_mem_gb"])
    except Exception as e: print(f, "failed", e)
PY
