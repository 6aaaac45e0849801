#!/bin/bash
# persistent decode attention with TWO workgroups per CU (2-slot rings): bit-identity test, A/B of the per-layer launch and of the decode iteration
mkdir -p gpurun_out/r04
python3 -m pytest "tests/test_gpu_kernels.py" -k "decode_attention_persistent" -q 2>&1 | tail -4
for mode in items persistent p2; do
  echo "== decode attention launch, ST_DECODE_ATTN=$mode"
  ST_DECODE_ATTN=$mode python3 tools/decode_attn_bench.py 43 8 1152 256 2>&1 | tail -4
  ST_DECODE_ATTN=$mode python3 tools/decode_attn_bench.py 64 8 1102 448 2>&1 | tail -4
  ST_DECODE_ATTN=$mode python3 tools/decode_attn_bench.py 64 8 1102 64 2>&1 | tail -4
  ST_DECODE_ATTN=$mode python3 tools/gen_flat.py 200 64 8 2>&1 | grep "^rows" | tail -1
  ST_DECODE_ATTN=$mode python3 tools/gen_flat.py 200 32 8 2>&1 | grep "^rows" | tail -1
done
