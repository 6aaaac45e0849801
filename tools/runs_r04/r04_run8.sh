# round 4, call 8: rollout-recorded log-probs — parity test, then the bench in both modes (same box)
mkdir -p gpurun_out/r04
python3 -m pytest tests/test_gpu_rollout.py tests/test_gpu_kernels.py -k "rollout_recorded or decode_attention_persistent or prompt_cache or decode_fused" -q -s > gpurun_out/r04/tests_run8.log 2>&1
grep -n "max |d|\|measured\|passed\|failed" gpurun_out/r04/tests_run8.log | tail -12
python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r04/bench_b_recompute.json 2> gpurun_out/r04/bench_b.err; tail -c 300 gpurun_out/r04/bench_b.err
python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --old-from-rollout > gpurun_out/r04/bench_b_old_from_rollout.json 2> gpurun_out/r04/bench_b2.err; tail -c 300 gpurun_out/r04/bench_b2.err
python3 -c "
import json
for f in ('bench_b_recompute', 'bench_b_old_from_rollout'):
    d = json.load(open('gpurun_out/r04/' + f + '.json'))
    print(f, round(d['value'], 3), {k: round(v, 3) for k, v in d['timing_s'].items()}, round(d['roofline_decode']['ms_per_iteration'], 3), d['old_log_probs'][:40], round(d['roofline']['frac'], 4))"
