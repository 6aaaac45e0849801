# per-kernel timing of the attention micro-benchmark (tools/attn_bench.py) under rocprofv3
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python3 tools/attn_bench.py ${1:-4} ${2:-1614} 2>&1 | tail -9
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/attnprof -o x -- python3 tools/attn_bench.py ${1:-4} ${2:-1614} > /tmp/attnprof.log 2>&1; tail -5 /tmp/attnprof.log; ls -R /tmp/attnprof | head
python3 - <<'PY'
import csv,glob
fs=glob.glob('/tmp/attnprof/**/*kernel_stats.csv',recursive=True)
print(fs)
for r in csv.DictReader(open(fs[0])):
    if 'attn' in r['Name']: print(r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3)
PY
