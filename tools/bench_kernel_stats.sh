cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r02n -o x -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r02_bench_prof_n.json 2> /tmp/prof_n.err
ls /tmp/prof_r02n | head
cp /tmp/prof_r02n/x_kernel_stats.csv gpurun_out/r02_bench_kernel_stats_n.csv
tail -c 400 gpurun_out/r02_bench_prof_n.json
