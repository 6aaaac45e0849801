#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "asm4 or pipeline_lengths" > gpurun_out/r03_gputests_4.log 2>&1; echo "pytest rc=$?"
tail -25 gpurun_out/r03_gputests_4.log
timeout 300 python tools/gemm_sustained.py 3 > gpurun_out/r03_gemm_sustained_b.log 2>&1; cat gpurun_out/r03_gemm_sustained_b.log
