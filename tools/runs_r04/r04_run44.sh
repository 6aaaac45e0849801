#!/bin/bash
# rocprofv3 kernel stats of the default and the fp8 bench on the final sources
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
rm -rf /tmp/prof_a /tmp/prof_b
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_a -o x -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r04/final_bench_prof.json 2> /tmp/prof_a.err
cp /tmp/prof_a/x_kernel_stats.csv gpurun_out/r04/final_bench_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o x -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --dtype fp8 --fp8-dgrad --fp8-wgrad > gpurun_out/r04/fp8_bench_prof.json 2> /tmp/prof_b.err
cp /tmp/prof_b/x_kernel_stats.csv gpurun_out/r04/fp8_bench_kernel_stats.csv
python3 -c "
import json
for f in ('final_bench_prof', 'fp8_bench_prof'):
    d = json.loads(open('gpurun_out/r04/' + f + '.json').read().strip().splitlines()[-1])
    print(f, d['value'], d['timing_s'], d['roofline']['avg_launch_ms'])"
