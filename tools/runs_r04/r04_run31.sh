#!/bin/bash
# what the box gives the CPU baseline: cgroup quota / throttling / load, and the small-op legs at 4 / 8 / 16 threads
mkdir -p gpurun_out/r04
python3 tools/cpu_probe.py 2>&1 | tee gpurun_out/r04/cpu_probe.txt
