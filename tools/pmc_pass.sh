#!/bin/bash
# One rocprofv3 PMC pass (kernel trace + counters only) over a command, summarised per kernel:
#   tools/pmc_pass.sh <tag> "<counters>" <python script and args...>      -> gpurun_out/pmc_<tag>.json
# (run from the repo root on the GPU box; the program sits directly after `--`, never behind env/bash)
set -e
tag=$1; counters=$2; shift 2
root=$(pwd)
export TMPDIR=/tmp
d=/tmp/pmc_$tag
rm -rf $d
(cd /tmp && PYTHONPATH=$root rocprofv3 --kernel-trace --pmc $counters --output-format csv -d $d -o x -- python3 $root/$1 "${@:2}" > $d.log 2>&1) || tail -5 $d.log
python3 tools/pmc_summarize.py $d gpurun_out/pmc_$tag.json
rm -rf $d
