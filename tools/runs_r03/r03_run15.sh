#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python tools/swiglu_ab.py > gpurun_out/r03_swiglu_ab.log 2>&1; cat gpurun_out/r03_swiglu_ab.log
