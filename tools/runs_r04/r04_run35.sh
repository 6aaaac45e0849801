#!/bin/bash
# whole GPU suite, log kept
mkdir -p gpurun_out/r04
timeout 2700 python -m pytest tests -q -m gpu > gpurun_out/r04/tests_final.log 2>&1
grep -E "passed|failed|error" gpurun_out/r04/tests_final.log | tail -5
