#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_e2e.py -x -q -m gpu -k "fp32_master" 2>&1 | tail -25
