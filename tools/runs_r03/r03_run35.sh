#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
bash tools/bench_traffic.sh 2>&1 | tail -12
