# kernel durations of the ViT window attention probe (rocprofv3 --kernel-trace --stats; program directly after `--`)
#   bash tools/vit_window_prof.sh <images>
n=${1:-4}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
rm -rf /tmp/prof_win_$n
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_win_$n -o x -- python3 tools/vit_window_probe.py $n > /tmp/prof_win_$n.log 2>&1
python3 - <<PY
import csv
for r in csv.DictReader(open('/tmp/prof_win_$n/x_kernel_stats.csv')):
    if 'attn' in r['Name']:
        print(f"$n images  {r['Name'][:60]:60s} calls {int(r['Calls']):5d}  avg {float(r['AverageNs'])/1e3:8.1f} us  min {float(r['MinNs'])/1e3:8.1f}")
PY
