#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
(echo "== 32 tiles (32 CUs busy)"; python tools/epilogue_cost.py 1024 2048; echo "== 256 tiles (one synchronized round)"; python tools/epilogue_cost.py 4096 4096; echo "== 8 tiles"; python tools/epilogue_cost.py 512 1024) > gpurun_out/r03_epilogue_cost_small.log 2>&1; cat gpurun_out/r03_epilogue_cost_small.log
