# round 4, call 11: final sources — GPU suite, PMC traffic (training GEMM class + decode iteration), default bench line, rocprofv3 kernel stats
mkdir -p gpurun_out/r04
python3 -m pytest tests -m gpu -q > gpurun_out/r04/tests_full_b.log 2>&1
tail -3 gpurun_out/r04/tests_full_b.log
bash tools/bench_traffic.sh > gpurun_out/r04/bench_traffic.log 2>&1
tail -2 gpurun_out/r04/bench_traffic.log | cut -c1-700
cp gpurun_out/r04_gemm_traffic.json profiles/r04_gemm_traffic.json
python3 bench.py --steps 3 --warmup 1 > gpurun_out/r04/bench_final.json 2> gpurun_out/r04/bench_final.err; tail -c 400 gpurun_out/r04/bench_final.err
python3 -c "
import json; d = json.load(open('gpurun_out/r04/bench_final.json'))
print(d['value'], d['timing_s'], d['roofline']['frac'], d['roofline'].get('traffic'), d['roofline_decode']['ms_per_iteration'], d['roofline_decode']['traffic'], d['cpu_baseline']['value'])"
bash tools/bench_kernel_stats.sh r04 2>&1 | tail -3
