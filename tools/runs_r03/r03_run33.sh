#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
bash tools/gen_flat_prof.sh 100 64 8 > gpurun_out/r03_gen_flat_512.log 2>&1; cat gpurun_out/r03_gen_flat_512.log | cut -c1-175
bash tools/gen_flat_prof.sh 100 32 8 > gpurun_out/r03_gen_flat_256.log 2>&1; cat gpurun_out/r03_gen_flat_256.log | cut -c1-175
