# round 4, call 4: new parity tests (re-conditioned trajectory), DP tests on the pooled reducer, bench sanity + cpu_baseline timing
mkdir -p gpurun_out/r04
python3 -m pytest tests/test_gpu_trajectory.py tests/test_gpu_depth.py tests/test_gpu_dp.py -q -s > gpurun_out/r04/new_parity_tests_b.log 2>&1
grep -n "^call\|measured\|accumulated\|weights:\|full-depth\|passed\|failed" gpurun_out/r04/new_parity_tests_b.log | tail -50
python3 bench.py --steps 2 --warmup 1 > gpurun_out/r04/bench_a.json 2> gpurun_out/r04/bench_a.err
tail -c 1500 gpurun_out/r04/bench_a.err
python3 -c "
import json; d = json.load(open('gpurun_out/r04/bench_a.json'))
print(d['value'], d['timing_s'], d['roofline_decode']['ms_per_iteration'], d['cpu_baseline']['value'], d['cpu_baseline']['value_excl_generation_and_adamw'], d['cpu_baseline']['cores'])
print(d['cpu_baseline']['measured_s']); print(d['cpu_baseline']['spread_min_max_s'])"
