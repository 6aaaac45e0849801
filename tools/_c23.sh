mkdir -p gpurun_out/r06
bash tools/gemm_group_probe.sh > gpurun_out/r06/c23_group.log 2>&1
B="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fp8-leg --no-telemetry"
$B > gpurun_out/r06/c23_rl_on.json 2>/dev/null
$B --no-recompute-light > gpurun_out/r06/c23_rl_off.json 2>/dev/null
$B > gpurun_out/r06/c23_rl_on2.json 2>/dev/null
$B --no-recompute-light > gpurun_out/r06/c23_rl_off2.json 2>/dev/null
cat gpurun_out/r06/gemm_group_probe.txt
