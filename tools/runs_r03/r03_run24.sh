#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests/test_gpu_token_budget.py -x -q -m gpu -s 2>&1 | grep -E "measured|passed|failed|Assert" | head -20
