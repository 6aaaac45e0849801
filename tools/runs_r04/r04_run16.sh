#!/bin/bash
# bench: bf16 vs --dtype fp8 (4-wave fp8 tile) vs fp8 on the 8-wave tile, same box
mkdir -p gpurun_out/r04
for mode in bf16 fp8; do
  timeout 900 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --dtype $mode > gpurun_out/r04/bench_c_$mode.json 2> gpurun_out/r04/bench_c_$mode.err
  python3 - <<PY
import json
d = json.loads(open("gpurun_out/r04/bench_c_$mode.json").read().strip().splitlines()[-1])
print("$mode", d["value"], d["timing_s"], d.get("roofline", {}).get("achieved"))
for c in d.get("roofline_classes", []):
    if "fp8" in c["kernel"] or "mxfp8" in c["kernel"]: print("   ", c["kernel"][:60], c["achieved"], c["frac"])
PY
done
ST_FP8_TILE=8 timeout 900 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --dtype fp8 > gpurun_out/r04/bench_c_fp8_8wave.json 2> gpurun_out/r04/bench_c_fp8_8wave.err
python3 -c "
import json
d = json.loads(open('gpurun_out/r04/bench_c_fp8_8wave.json').read().strip().splitlines()[-1])
print('fp8 8-wave', d['value'], d['timing_s'])"
