# per-(kernel, grid) times of one fixed-batch decode phase (qkv and down share a tile kernel: the grid tells them apart):
#   bash tools/gen_flat_trace.sh [LEN] [prompts] [G]
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf /tmp/genflat_tr
rocprofv3 --kernel-trace --output-format csv -d /tmp/genflat_tr -o x -- python3 tools/gen_flat.py ${1:-200} ${2:-64} ${3:-8} > /tmp/genflat_tr.log 2>&1; grep "^rows" /tmp/genflat_tr.log
python3 - <<'PY'
import csv, collections, glob
f = glob.glob('/tmp/genflat_tr/**/x_kernel_trace.csv', recursive=True)[0]
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    k = (r['Kernel_Name'][:90], r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size', ''), r.get('Workgroup_Size_X', r.get('Workgroup_Size', '')))
    a = agg[k]; a[0] += 1; a[1] += float(r['End_Timestamp']) - float(r['Start_Timestamp'])
tot = sum(a[1] for a in agg.values())
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:26]:
    g = int(k[1]) // max(1, int(k[2])) if k[1] and k[2] else 0
    print(f"{k[0]:90s} wgs {g:6d} x{k[2]:>4s} calls {a[0]:6d} avg {a[1] / a[0] / 1e3:8.1f}us {100 * a[1] / tot:5.1f}%")
PY
