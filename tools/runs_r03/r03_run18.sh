#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O3 tools/micro/store_rate.hip -o /tmp/store_rate 2>/dev/null && /tmp/store_rate > gpurun_out/r03_store_rate.log 2>&1; cat gpurun_out/r03_store_rate.log
