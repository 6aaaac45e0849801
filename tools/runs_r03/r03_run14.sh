#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python tools/gemm_shapes.py 21504 > gpurun_out/r03_gemm_shapes.log 2>&1; cat gpurun_out/r03_gemm_shapes.log
for v in 40 lib; do
bash tools/pmc_pass.sh g_${v}_sq "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" tools/gemm_one.py $v 28672 37888 3584
bash tools/pmc_pass.sh g_${v}_sq2 "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA" tools/gemm_one.py $v 28672 37888 3584
bash tools/pmc_pass.sh g_${v}_sq3 "SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" tools/gemm_one.py $v 28672 37888 3584
done
python - <<'PY'
import json
for v in ("40","lib"):
    for f in ("sq","sq2","sq3"):
        d=json.load(open(f"gpurun_out/pmc_g_{v}_{f}.json"))
        for k,val in d.items():
            if "gemm" in k or "Cijk" in k: print(v, f, k[:40], {a:(round(b/val["dispatches"]/1e6,2) if a!="dispatches" else b) for a,b in val.items()})
PY
