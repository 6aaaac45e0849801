#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for npr in 2 8 16 24 32 40 48 56 64; do python tools/gen_flat.py 120 $npr 8 2>&1 | grep "^rows" | tail -1; done > gpurun_out/r03_decode_curve.log
cat gpurun_out/r03_decode_curve.log
