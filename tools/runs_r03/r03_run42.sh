#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for tg in 24576 32768 40960; do
  ST_TOKENS_GRAD=$tg timeout 900 python bench.py --no-cpu-baseline > gpurun_out/r03_bench_tg$tg.json 2> gpurun_out/r03_bench_tg$tg.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/r03_bench_tg$tg.json").read().strip().splitlines()[-1])
print($tg, round(d["value"], 3), {k: round(v, 2) for k, v in d["timing_s"].items()}, "peak", round(d["peak_mem_gb"], 1), "reserved", d["reserved_gb_after_each_step"], d["passes_per_step"], round(d["roofline"]["frac"], 4))
PY
done
