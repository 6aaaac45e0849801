mkdir -p gpurun_out/r06
python3 -m cProfile -o /tmp/bench.prof bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-fp8-leg --no-telemetry > /tmp/b.json 2>/tmp/b.err
python3 - <<'PY' > gpurun_out/r06/c11_hostprof.txt
import pstats
p = pstats.Stats('/tmp/bench.prof')
p.sort_stats('tottime').print_stats(45)
p.sort_stats('cumtime').print_stats(70)
PY
head -120 gpurun_out/r06/c11_hostprof.txt
