#!/bin/bash
# rocprofv3 kernel stats of the fp8 bench (forward + dgrad + wgrad in MX-fp8), default workload
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
rm -rf /tmp/prof_fp8
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_fp8 -o x -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --dtype fp8 --fp8-dgrad --fp8-wgrad > gpurun_out/r04/fp8_bench_prof.json 2> /tmp/prof_fp8.err
cp /tmp/prof_fp8/x_kernel_stats.csv gpurun_out/r04/fp8_bench_kernel_stats.csv
tail -c 400 gpurun_out/r04/fp8_bench_prof.json
