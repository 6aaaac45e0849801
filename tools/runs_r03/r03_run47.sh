#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py tests/test_gpu_model.py -x -q -m gpu -k "vit or fullsize or model or tiny or patch or image" 2>&1 | tail -6
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/r03_bench_i.json 2> gpurun_out/r03_bench_i.err; python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03_bench_i.json").read().strip().splitlines()[-1])
print(d["value"], {k: round(v, 2) for k, v in d["timing_s"].items()}, d["peak_mem_gb"])
for c in d["roofline_classes"]:
    print("  ", c["kernel"][:70], round(c["achieved"], 1), c["unit"], round(c["frac"], 4))
PY
timeout 900 python bench.py --dtype fp8 --rollouts 16 --prompts-per-gpu 32 --image 896x896 --no-cpu-baseline > gpurun_out/r03_bench_cfg5b.json 2> gpurun_out/r03_bench_cfg5b.err; python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03_bench_cfg5b.json").read().strip().splitlines()[-1])
print("cfg5", d["value"], {k: round(v, 2) for k, v in d["timing_s"].items()}, d["peak_mem_gb"])
for c in d["roofline_classes"]:
    print("  ", c["kernel"][:70], round(c["achieved"], 1), c["unit"], round(c["frac"], 4))
PY
