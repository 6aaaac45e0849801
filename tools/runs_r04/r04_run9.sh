# round 4, call 9: 257..512-row gate/up on the 4-wave tile with a K-split SwiGLU tail — parity, GEMM timing, decode iteration A/B
mkdir -p gpurun_out/r04
python3 -m pytest tests/test_gpu_production_shapes.py tests/test_gpu_kernels.py tests/test_gpu_rollout.py -q -x > gpurun_out/r04/tests_run9.log 2>&1
tail -4 gpurun_out/r04/tests_run9.log
python3 tools/decode512_probe.py 320 512 2>&1 | grep "gate/up"
for m in 1 0; do
  echo "== ST_DECODE_GU_ASM4=$m"
  ST_DECODE_GU_ASM4=$m python3 tools/gen_flat.py 200 64 8 2>&1 | grep "^rows" | tail -1
  ST_DECODE_GU_ASM4=$m python3 tools/gen_flat.py 200 44 8 2>&1 | grep "^rows" | tail -1
done
