#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for tn in 65536 98304 131072; do
ST_TOKENS_NOGRAD=$tn ST_FUSE_EXPERIENCE=32 timeout 900 python bench.py --no-cpu-baseline > gpurun_out/r03_bench_tn$tn.json 2> gpurun_out/r03_bench_tn$tn.err; python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r03_bench_tn$tn.json").read().strip().splitlines()[-1])
    print("nograd tokens $tn", round(d["value"], 3), {k: round(v, 2) for k, v in d["timing_s"].items()}, d["peak_mem_gb"], d["passes_per_step"])
except Exception as e:
    print("$tn failed", e); print(open("gpurun_out/r03_bench_tn$tn.err").read()[-600:])
PY
done
