#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in 40 lib; do
bash tools/pmc_pass.sh f_${v}_sq "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" tools/gemm_one.py $v 28672 37888 3584
bash tools/pmc_pass.sh f_${v}_sq2 "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA" tools/gemm_one.py $v 28672 37888 3584
done
bash tools/pmc_pass.sh f_tn_sq "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" tools/gemm_one.py tn 37888 3584 21504
bash tools/pmc_pass.sh f_tn_sq2 "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA" tools/gemm_one.py tn 37888 3584 21504
bash tools/pmc_pass.sh f_tn_l2 "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" tools/gemm_one.py tn 37888 3584 21504
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/pmc_f_*.json")):
    d = json.load(open(f))
    for k, val in d.items():
        if "gemm_nt4" in k or "Cijk" in k: print(f.split("/")[-1], k[:48], {a: (round(b / val["dispatches"] / 1e6, 2) if a != "dispatches" else b) for a, b in val.items()})
PY
