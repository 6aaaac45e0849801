#!/bin/bash
# final sources: GPU suite, smoke(), the driver's bench command (20 steps + 5 warm-up), rocprofv3 kernel stats of the default and the fp8 bench
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -4
timeout 600 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 1500 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04/bench_final_20.json 2> gpurun_out/r04/bench_final_20.err
python3 -c "
import json
d = json.loads(open('gpurun_out/r04/bench_final_20.json').read().strip().splitlines()[-1])
print('20 steps:', d['value'], d['timing_s'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['cpu_baseline']['value'], d['roofline_decode']['ms_per_iteration'], d['reserved_gb_after_each_step'][-3:])"
rm -rf /tmp/prof_a /tmp/prof_b
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_a -o x -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r04/final_bench_prof.json 2> /tmp/prof_a.err
cp /tmp/prof_a/x_kernel_stats.csv gpurun_out/r04/final_bench_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o x -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --dtype fp8 --fp8-dgrad --fp8-wgrad > gpurun_out/r04/fp8_bench_prof.json 2> /tmp/prof_b.err
cp /tmp/prof_b/x_kernel_stats.csv gpurun_out/r04/fp8_bench_kernel_stats.csv
echo done
