#!/bin/bash
# the 4-wave fp8 tile (gemm_mx4.hip): parity on both tiles, then throughput next to the 8-wave tile and the bf16 kernel
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_fp8.py -x -q -k "gemm_vs or quantiser" 2>&1 | tail -15
timeout 600 python tools/gemm_fp8_bench.py 10496 2>&1 | tee gpurun_out/r04/fp8_gemm_bench.txt
