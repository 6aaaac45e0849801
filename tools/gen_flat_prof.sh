# per-kernel times of one fixed-batch decode phase:  bash tools/gen_flat_prof.sh [LEN] [prompts] [G]
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/genflat -o x -- python3 tools/gen_flat.py ${1:-200} ${2:-64} ${3:-8} > /tmp/genflat.log 2>&1; grep "^rows" /tmp/genflat.log
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('/tmp/genflat/x_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:22]:
    print(f"{r['Name'][:100]:100s} {int(r['Calls']):7d} {float(r['AverageNs'])/1e3:8.1f}us {100*float(r['TotalDurationNs'])/tot:5.1f}%")
PY
