#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_train_options.py tests/test_gpu_production_shapes.py -x -q -m gpu -s -k "master or training or adamw" 2>&1 | grep -E "step [0-9] |passed|failed|Error|assert" | head -40
