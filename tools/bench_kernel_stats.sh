# rocprofv3 --kernel-trace --stats over the default bench command; the summary lands in gpurun_out/<tag>_bench_kernel_stats.csv
#   bash tools/bench_kernel_stats.sh <tag>        (from the repo root on the GPU box; the program sits directly after `--`)
tag=${1:-r03}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o x -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fp8-leg --no-telemetry > gpurun_out/${tag}_bench_prof.json 2> /tmp/prof_$tag.err
ls /tmp/prof_$tag | head
cp /tmp/prof_$tag/x_kernel_stats.csv gpurun_out/${tag}_bench_kernel_stats.csv
tail -c 300 gpurun_out/${tag}_bench_prof.json
