mkdir -p gpurun_out/r06
python3 tools/parity_probe.py 16 --json gpurun_out/r06/c4_parity.json > gpurun_out/r06/c4_parity.txt 2>&1
B="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fp8-leg"
$B --no-telemetry > gpurun_out/r06/c4_bench_notele.json 2> gpurun_out/r06/c4_bench_notele.err
$B > gpurun_out/r06/c4_bench_tele.json 2> gpurun_out/r06/c4_bench_tele.err
$B --no-telemetry --fuse-micro-batches 12 --tokens-grad 32768 > gpurun_out/r06/c4_bench_f12.json 2> gpurun_out/r06/c4_bench_f12.err
$B --no-telemetry --fuse-micro-batches 16 --tokens-grad 45056 > gpurun_out/r06/c4_bench_f16.json 2> gpurun_out/r06/c4_bench_f16.err
tail -5 gpurun_out/r06/c4_parity.txt
