#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python tools/gemm_ksweep.py debug > gpurun_out/r03_gemm_ksweep_dbg2.log 2>&1; cat gpurun_out/r03_gemm_ksweep_dbg2.log
