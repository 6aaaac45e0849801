#!/bin/bash
# more reference micro-batches per update pass: --fuse-micro-batches 16 with a 49152-token budget vs the default (8 / 24576), same box
mkdir -p gpurun_out/r04
for cfg in "8 24576" "16 49152"; do
  set -- $cfg
  ST_TOKENS_GRAD=$2 timeout 900 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --fuse-micro-batches $1 > gpurun_out/r04/bench_fm_$1.json 2> gpurun_out/r04/bench_fm_$1.err
  python3 -c "
import json
d = json.loads(open('gpurun_out/r04/bench_fm_$1.json').read().strip().splitlines()[-1])
print('$cfg', d['value'], d['timing_s']['update_actor'], d['passes_per_step'], d['peak_mem_gb'], d['peak_reserved_gb'], d['roofline']['frac'])" || tail -3 gpurun_out/r04/bench_fm_$1.err
done
