#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
bash tools/bench_kernel_stats.sh r03_k
head -12 gpurun_out/r03_k_bench_kernel_stats.csv | cut -c1-160
