# PMC counters of the ViT window attention probe, one pass per counter group (kernel trace + counters only)
#   bash tools/vit_window_pmc.sh <images> <counter> [<counter> ...]
n=$1; shift
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
for c in "$@"; do
  d=/tmp/pmc_win_$c
  rm -rf $d
  timeout 300 rocprofv3 --kernel-trace --pmc $c --kernel-include-regex "attn" --output-format csv -d $d -o x -- python3 tools/vit_window_probe.py $n > /tmp/pmc_win_$c.log 2>&1 || tail -3 /tmp/pmc_win_$c.log
  python3 tools/pmc_summarize.py $d gpurun_out/r05/pmc_win_${n}_$c.json attn
done
python3 - <<PY
import json, glob
for f in sorted(glob.glob('gpurun_out/r05/pmc_win_${n}_*.json')):
    for k, v in json.load(open(f)).items():
        n_ = v.pop('dispatches')
        print(k[:40], {c: round(x / n_ / 1e6, 3) for c, x in v.items()}, '(per launch, 1e6 units)')
PY
