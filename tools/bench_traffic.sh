#!/bin/bash
# HBM traffic of the bench's dominant kernels, rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE separately; kernel trace + counters only):
#   training GEMM class: `bench.py --steps 1 --warmup 0 --no-cpu-baseline` with counter collection restricted to the GEMM kernels
#       (--kernel-include-regex; counters on every kernel of the run serialise ~600k launches and take > 30 min per pass);
#   decode iteration: tools/gen_flat.py 6 64 8 (64 prompts x 8 rollouts, 512 live rows, 5 decode iterations, twice), every kernel.
# Summarised per kernel, then profiles-ready JSON (tools/make_traffic_json.py).  Run from the repo root on the GPU box; the program
# sits directly after `--`.
cd /tmp && export TMPDIR=/tmp
root=$GRAFT_REPO_ROOT
cd $root
mkdir -p gpurun_out
RX='gemm_nt4_kernel'      # (round 6: with the finish / 128x128 / 'gemm_tile_kernel<256, 256' kernels in the list the counter pass segfaulted inside the bench process, 6 of 6 times on three boxes; the 4-wave tile alone is > 95 % of the class's time)
for c in FETCH_SIZE WRITE_SIZE; do
  d=/tmp/pmc_bench_$c
  rm -rf $d
  (cd /tmp && PYTHONPATH=$root timeout 700 rocprofv3 --kernel-trace --pmc $c --kernel-include-regex "$RX" --output-format csv -d $d -o x -- python3 $root/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-fp8-leg --no-telemetry > $root/gpurun_out/bench_traffic_$c.json 2> /tmp/bench_traffic_$c.err) || tail -5 /tmp/bench_traffic_$c.err
  python3 tools/pmc_summarize.py $d gpurun_out/pmc_bench_$c.json
  rm -rf $d
  d=/tmp/pmc_dec_$c
  rm -rf $d
  timeout 420 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -o x -- python3 tools/gen_flat.py 6 64 8 > gpurun_out/dec_traffic_$c.log 2>&1 || tail -5 gpurun_out/dec_traffic_$c.log
  python3 tools/pmc_summarize.py $d gpurun_out/pmc_dec_$c.json
  rm -rf $d
done
python3 tools/make_traffic_json.py gpurun_out/pmc_bench_FETCH_SIZE.json gpurun_out/pmc_bench_WRITE_SIZE.json gpurun_out/bench_traffic_FETCH_SIZE.json gpurun_out/r05_gemm_traffic.json gpurun_out/pmc_dec_FETCH_SIZE.json gpurun_out/pmc_dec_WRITE_SIZE.json 10
