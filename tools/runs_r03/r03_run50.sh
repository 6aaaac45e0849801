#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  d=/tmp/pmc_dec_$c; rm -rf $d
  timeout 420 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -o x -- python3 tools/gen_flat.py 6 64 8 > gpurun_out/dec_traffic_$c.log 2>&1; echo "rc=$?"; grep "^rows" gpurun_out/dec_traffic_$c.log | tail -2; tail -3 gpurun_out/dec_traffic_$c.log | cut -c1-200
  python3 tools/pmc_summarize.py $d gpurun_out/pmc_dec_$c.json
  rm -rf $d
done
