# round 4, call 3: the two new parity tests (trajectory over 3 update_actor calls; full-depth 7B log-probs)
mkdir -p gpurun_out/r04
python3 -m pytest tests/test_gpu_trajectory.py tests/test_gpu_depth.py -q -s > gpurun_out/r04/new_parity_tests.log 2>&1
tail -40 gpurun_out/r04/new_parity_tests.log
