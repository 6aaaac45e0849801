#!/bin/bash
# packed-token budget of an update pass: same-box A/B of ST_TOKENS_GRAD (default 24576)
mkdir -p gpurun_out/r04
for tg in 24576 32768 40960; do
  ST_TOKENS_GRAD=$tg timeout 900 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r04/bench_tg_$tg.json 2> gpurun_out/r04/bench_tg_$tg.err
  python3 -c "
import json
d = json.loads(open('gpurun_out/r04/bench_tg_$tg.json').read().strip().splitlines()[-1])
print($tg, d['value'], d['timing_s']['update_actor'], d['passes_per_step'], d['peak_mem_gb'], d['peak_reserved_gb'], d['roofline']['frac'])"
done
