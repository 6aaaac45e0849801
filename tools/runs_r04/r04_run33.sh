#!/bin/bash
# the critic (adv_estimator = gae): kernels, values / gradients vs the oracle, update_critic loop, PPO end to end
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_critic.py -x -q -s 2>&1 | grep -v "^$" | tail -25
timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_model.py tests/test_gpu_trajectory.py -x -q 2>&1 | tail -4
