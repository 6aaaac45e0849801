#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for v in 40 lib; do
  bash tools/pmc_pass.sh l2_${v}_hit "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" tools/gemm_one.py $v 28672 37888 3584
  bash tools/pmc_pass.sh l2_${v}_ea "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum" tools/gemm_one.py $v 28672 37888 3584
  bash tools/pmc_pass.sh l2_${v}_tcp "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" tools/gemm_one.py $v 28672 37888 3584
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/pmc_l2_*.json")):
    d = json.load(open(f))
    print(f)
    for k in d if isinstance(d, list) else d.get("kernels", d):
        print("  ", json.dumps(k)[:400] if not isinstance(k, str) else (k, json.dumps(d[k])[:400]))
PY
