#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -q --durations=8 > gpurun_out/r03_gputests_12.log 2>&1; echo "pytest rc=$?"
tail -25 gpurun_out/r03_gputests_12.log
