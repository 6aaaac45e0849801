#!/bin/bash
# the 4-wave fp8 tile's K loop by elimination
mkdir -p gpurun_out/r04
timeout 300 python tools/mx4_ksweep.py 2>&1 | tee gpurun_out/r04/mx4_ksweep.txt
timeout 300 python tools/mx4_ksweep.py 16384 4608 3584 2>&1 | tee -a gpurun_out/r04/mx4_ksweep.txt
timeout 300 python tools/gemm_ksweep.py debug 2>&1 | tail -12 | tee -a gpurun_out/r04/mx4_ksweep.txt
