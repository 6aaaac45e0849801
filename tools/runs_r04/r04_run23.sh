#!/bin/bash
# the 4-wave bf16 tile on the 32-cycle MFMA: results and throughput against the production tile
mkdir -p gpurun_out/r04
timeout 600 python tools/asm4w_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04/asm4w_bench.txt
