#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
bash tools/bench_kernel_stats.sh r03_g
head -40 gpurun_out/r03_g_bench_kernel_stats.csv | cut -c1-200
