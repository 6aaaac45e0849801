#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python tools/decode_attn_bench.py > gpurun_out/r03_decode_attn_bench.log 2>&1; tail -12 gpurun_out/r03_decode_attn_bench.log
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_rollout.py tests/test_gpu_production_shapes.py -x -q -m gpu -k "attn or decode or rollout or generate" 2>&1 | tail -4
for npr in 8 32 48 64; do python tools/gen_flat.py 120 $npr 8 2>&1 | grep "^rows" | tail -1; done
