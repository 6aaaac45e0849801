mkdir -p gpurun_out/r06
(timeout 900 python3 -m pytest tests/test_gpu_rollout.py -x -q -s -k "fp8_mode_runs" 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -12) > gpurun_out/r06/c20_tests.txt
cat gpurun_out/r06/c20_tests.txt
F="python3 bench.py --dtype fp8 --fp8-dgrad --fp8-wgrad --rollouts 16 --prompts-per-gpu 32 --image 896x896 --steps 2 --warmup 1 --no-cpu-baseline --no-fp8-leg --no-telemetry"
ST_FP8_DECODE=0 $F > gpurun_out/r06/c20_cfg5_off.json 2>/dev/null
$F > gpurun_out/r06/c20_cfg5_on.json 2>/dev/null
