# round 4, call 5: trajectory bounds, chunked rollout, real tokenizer/processor e2e, then the shipped scripts' worst-case shape
mkdir -p gpurun_out/r04
python3 -m pytest tests/test_gpu_trajectory.py "tests/test_gpu_rollout.py::test_prompt_chunked_rollout_matches_the_unchunked_call_and_overflow_fails_cleanly" "tests/test_gpu_e2e.py::test_main_with_a_real_tokenizer_and_processor" -q -s > gpurun_out/r04/tests_run5.log 2>&1
grep -n "measured\|zero-gradient\|worst of the other\|weights:\|passed\|failed\|Error" gpurun_out/r04/tests_run5.log | tail -30
timeout 2000 python3 bench.py --worst-case > gpurun_out/r04/bench_worst_case.json 2> gpurun_out/r04/bench_worst_case.err
echo "worst-case exit: $?"
tail -c 1200 gpurun_out/r04/bench_worst_case.err
python3 -c "
import json; d = json.load(open('gpurun_out/r04/bench_worst_case.json'))
for k in ('value','error','ms_per_step','timing_s','peak_mem_gb','peak_reserved_gb','passes_per_step','rollout_prompt_chunks','prompt_cache_hit','prompt_tokens','unrelated_rows_probe'):
    print(k, d.get(k))
print(d.get('roofline_decode', {}) and {k: d['roofline_decode'][k] for k in ('ms_per_iteration','mean_rows_per_iteration','iterations')})"
