# round 4, call 6: persistent decode attention — bit-identity / parity tests, A/B of the per-layer launch and of the decode iteration
mkdir -p gpurun_out/r04
python3 -m pytest "tests/test_gpu_kernels.py" -k "decode_attention_persistent" -q -s > gpurun_out/r04/tests_run6a.log 2>&1
tail -5 gpurun_out/r04/tests_run6a.log
python3 -m pytest tests/test_gpu_rollout.py tests/test_gpu_production_shapes.py "tests/test_gpu_e2e.py::test_main_with_a_real_tokenizer_and_processor" -q -x > gpurun_out/r04/tests_run6b.log 2>&1
tail -5 gpurun_out/r04/tests_run6b.log
for mode in items persistent; do
  echo "== decode attention launch, ST_DECODE_ATTN=$mode"
  ST_DECODE_ATTN=$mode python3 tools/decode_attn_bench.py 43 8 1152 256 2>&1 | tail -4
  ST_DECODE_ATTN=$mode python3 tools/decode_attn_bench.py 64 8 1102 448 2>&1 | tail -4
  ST_DECODE_ATTN=$mode python3 tools/gen_flat.py 200 64 8 2>&1 | grep "^rows" | tail -1
  ST_DECODE_ATTN=$mode python3 tools/gen_flat.py 200 32 8 2>&1 | grep "^rows" | tail -1
  ST_DECODE_ATTN=$mode python3 tools/gen_flat.py 200 8 8 2>&1 | grep "^rows" | tail -1
done
