#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python tools/gemm_km_fuzz.py 1 120 2>&1 | tail -8
