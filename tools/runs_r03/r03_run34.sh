#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python tools/decode_overhead.py > gpurun_out/r03_decode_overhead.log 2>&1; cat gpurun_out/r03_decode_overhead.log
