mkdir -p gpurun_out/r06
(timeout 900 python3 -m pytest tests/test_gpu_kernels.py -x -q -k "rmsnorm" 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -15) > gpurun_out/r06/c14_tests.txt
(timeout 900 python3 -m pytest tests/test_gpu_model.py tests/test_gpu_trajectory.py tests/test_gpu_fullsize.py -x -q 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -8) >> gpurun_out/r06/c14_tests.txt
B="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fp8-leg --no-telemetry"
ST_RMSNORM_BWD_FUSED=0 $B > gpurun_out/r06/c14_off.json 2>/dev/null
$B > gpurun_out/r06/c14_on.json 2>/dev/null
cat gpurun_out/r06/c14_tests.txt
