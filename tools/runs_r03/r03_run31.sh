#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_production_shapes.py -x -q -m gpu -k "nn_tn or asm4 or training" > gpurun_out/r03_gputests_31.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r03_gputests_31.log
timeout 600 python tools/gemm_shapes.py 21504 > gpurun_out/r03_gemm_shapes5.log 2>&1; tail -6 gpurun_out/r03_gemm_shapes5.log
