# round 4, call 12: two ranks on one GPU through bench.py, the learning test, the opt-in plan-40 gate/up check
mkdir -p gpurun_out/r04
python3 -m pytest "tests/test_gpu_e2e.py::test_bench_two_ranks_on_one_gpu_runs_the_real_multi_rank_step" "tests/test_gpu_e2e.py::test_grpo_loop_learns_a_dense_synthetic_reward" "tests/test_gpu_production_shapes.py::test_decode_shaped_gemms_at_7b_shapes_vs_fp32" -q -s > gpurun_out/r04/tests_run12.log 2>&1
grep -n "share of sampled\|passed\|failed\|^E " gpurun_out/r04/tests_run12.log | head -20
