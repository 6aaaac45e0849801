#!/bin/bash
# HBM traffic of the bench's kernels: rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE separately, kernel trace + counters only) over
# `bench.py --steps 1 --warmup 0`, summarised per kernel, then profiles-ready JSON (tools/make_traffic_json.py).
# Run from the repo root on the GPU box; the program sits directly after `--`.
set -x
cd /tmp && export TMPDIR=/tmp
root=$GRAFT_REPO_ROOT
cd $root
mkdir -p gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  d=/tmp/pmc_bench_$c
  rm -rf $d
  (cd /tmp && PYTHONPATH=$root rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -o x -- python3 $root/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $root/gpurun_out/bench_traffic_$c.json 2> /tmp/bench_traffic_$c.err) || tail -5 /tmp/bench_traffic_$c.err
  python3 tools/pmc_summarize.py $d gpurun_out/pmc_bench_$c.json
  rm -rf $d
done
python3 tools/make_traffic_json.py gpurun_out/pmc_bench_FETCH_SIZE.json gpurun_out/pmc_bench_WRITE_SIZE.json gpurun_out/bench_traffic_FETCH_SIZE.json gpurun_out/r03_gemm_traffic.json
cat gpurun_out/r03_gemm_traffic.json
