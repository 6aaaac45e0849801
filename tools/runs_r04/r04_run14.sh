#!/bin/bash
# operand / scale layout of the K = 64 block-scaled fp8 MFMA (the 4-wave fp8 tile is built on it)
mkdir -p gpurun_out/r04
./tools/probes/mx_layout_probe32 > gpurun_out/r04/mx_layout_probe32.txt 2>&1
cat gpurun_out/r04/mx_layout_probe32.txt
