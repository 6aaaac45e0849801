#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_production_shapes.py -x -q -m gpu -k "swiglu or asm4 or training" 2>&1 | tail -4
timeout 600 python tools/swiglu_ab.py > gpurun_out/r03_swiglu_ab4.log 2>&1; cat gpurun_out/r03_swiglu_ab4.log
