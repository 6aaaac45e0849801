#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py tests/test_gpu_model.py -x -q -m gpu -k "vit or fullsize or model or tiny or patch or image" 2>&1 | tail -3
