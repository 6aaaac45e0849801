mkdir -p gpurun_out/r06
B="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fp8-leg --no-telemetry"
ST_PINNED_H2D=0 ST_PLAN_CACHE=0 $B > gpurun_out/r06/c13_old.json 2>/dev/null
$B > gpurun_out/r06/c13_new.json 2>/dev/null
ST_PINNED_H2D=0 ST_PLAN_CACHE=0 $B > gpurun_out/r06/c13_old2.json 2>/dev/null
$B > gpurun_out/r06/c13_new2.json 2>/dev/null
