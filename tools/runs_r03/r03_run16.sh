#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python tools/epilogue_cost.py > gpurun_out/r03_epilogue_cost.log 2>&1; cat gpurun_out/r03_epilogue_cost.log
