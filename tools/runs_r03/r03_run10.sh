#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "asm4 or tail_split" > gpurun_out/r03_gputests_10.log 2>&1; echo "pytest rc=$?"
tail -4 gpurun_out/r03_gputests_10.log
timeout 300 python tools/gemm_ksweep.py debug > gpurun_out/r03_gemm_ksweep_dbg3.log 2>&1; cat gpurun_out/r03_gemm_ksweep_dbg3.log
timeout 300 python tools/gemm_sustained.py 3 > gpurun_out/r03_gemm_sustained_f.log 2>&1; grep "randn" gpurun_out/r03_gemm_sustained_f.log
