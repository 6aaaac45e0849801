#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "nn_tn or asm4" > gpurun_out/r03_gputests_30.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/r03_gputests_30.log
timeout 600 python tools/gemm_shapes.py 21504 > gpurun_out/r03_gemm_shapes4.log 2>&1; cat gpurun_out/r03_gemm_shapes4.log
