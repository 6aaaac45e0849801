#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python tools/vit_attn_ab.py > gpurun_out/r03_vit_attn_ab.log 2>&1; cat gpurun_out/r03_vit_attn_ab.log
