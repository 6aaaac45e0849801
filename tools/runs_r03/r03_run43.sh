#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for cfg in "8 24576" "16 32768" "16 45056"; do
  set -- $cfg
  ST_TOKENS_GRAD=$2 timeout 900 python bench.py --no-cpu-baseline --fuse-micro-batches $1 > gpurun_out/r03_bench_f$1_$2.json 2> gpurun_out/r03_bench_f$1_$2.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r03_bench_f$1_$2.json").read().strip().splitlines()[-1])
    print("$1 $2", round(d["value"], 3), {k: round(v, 2) for k, v in d["timing_s"].items()}, "peak", round(d["peak_mem_gb"], 1), "reserved", d["reserved_gb_after_each_step"], d["passes_per_step"], round(d["roofline"]["frac"], 4))
except Exception as e:
    print("$1 $2 failed", e); print(open("gpurun_out/r03_bench_f$1_$2.err").read()[-800:])
PY
done
