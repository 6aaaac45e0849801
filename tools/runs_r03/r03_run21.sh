#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O3 tools/micro/store_pitch.hip -o /tmp/store_pitch 2>/dev/null && /tmp/store_pitch > gpurun_out/r03_store_pitch.log 2>&1; cat gpurun_out/r03_store_pitch.log
