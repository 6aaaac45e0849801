#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python tools/tile_trace.py > gpurun_out/r03_tile_trace.log 2>&1; cat gpurun_out/r03_tile_trace.log
