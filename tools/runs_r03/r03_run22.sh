#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python tools/epilogue_cost.py > gpurun_out/r03_epilogue_cost3.log 2>&1; cat gpurun_out/r03_epilogue_cost3.log
python tools/swiglu_ab.py > gpurun_out/r03_swiglu_ab3.log 2>&1; cat gpurun_out/r03_swiglu_ab3.log
python tools/gemm_shapes.py 21504 > gpurun_out/r03_gemm_shapes3.log 2>&1; cat gpurun_out/r03_gemm_shapes3.log
python -m pytest tests -x -q -m gpu > gpurun_out/r03_gputests_22.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r03_gputests_22.log
python bench.py > gpurun_out/r03_bench_f.json 2> gpurun_out/r03_bench_f.err; python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03_bench_f.json").read().strip().splitlines()[-1])
print(d["value"], d.get("phases_s") or d.get("config"), d["roofline"], d.get("roofline_decode"))
PY
