#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python tools/asm4_debug.py > gpurun_out/r03_asm4_debug.log 2>&1; cat gpurun_out/r03_asm4_debug.log
bash tools/pmc_pass.sh a4_sq "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" tools/gemm_one.py 40 28672 37888 3584
bash tools/pmc_pass.sh a4_sq2 "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_MFMA" tools/gemm_one.py 40 28672 37888 3584
bash tools/pmc_pass.sh a4_sq3 "SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_ACTIVE_INST_FLAT" tools/gemm_one.py 40 28672 37888 3584
python - <<'PY'
import json
for f in ("a4_sq","a4_sq2","a4_sq3"):
    d=json.load(open(f"gpurun_out/pmc_{f}.json"))
    for k,v in d.items():
        if "gemm" in k: print(f, k[:60], {a:(round(b/v["dispatches"]) if a!="dispatches" else b) for a,b in v.items()})
PY
