#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O3 tools/micro/cvt_check.hip -o /tmp/cvt_check 2>/dev/null && /tmp/cvt_check > gpurun_out/r03_cvt_check.log 2>&1; cat gpurun_out/r03_cvt_check.log
python -m pytest tests/test_gpu_token_budget.py -x -q -m gpu -s 2>&1 | grep -E "measured|passed|failed|Assert" | head
