#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "gemm" > gpurun_out/r03_gputests_19.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r03_gputests_19.log
python tools/epilogue_cost.py > gpurun_out/r03_epilogue_cost2.log 2>&1; cat gpurun_out/r03_epilogue_cost2.log
python tools/swiglu_ab.py > gpurun_out/r03_swiglu_ab2.log 2>&1; cat gpurun_out/r03_swiglu_ab2.log
python tools/gemm_shapes.py 21504 > gpurun_out/r03_gemm_shapes2.log 2>&1; cat gpurun_out/r03_gemm_shapes2.log
