#!/bin/bash
# how much of `gen` is neither prefill nor decode replays (graph capture, host bookkeeping, syncs)?
mkdir -p gpurun_out/r04
timeout 600 python tools/gen_phases.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04/gen_phases.txt
