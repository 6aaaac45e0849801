#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python tools/asm4_debug.py > gpurun_out/r03_asm4_debug.log 2>&1; tail -12 gpurun_out/r03_asm4_debug.log
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "asm4 or pipeline_lengths" > gpurun_out/r03_gputests_6.log 2>&1; echo "pytest rc=$?"
tail -12 gpurun_out/r03_gputests_6.log
timeout 300 python tools/gemm_sustained.py 3 > gpurun_out/r03_gemm_sustained_c.log 2>&1; cat gpurun_out/r03_gemm_sustained_c.log; timeout 300 python tools/gemm_ksweep.py > gpurun_out/r03_gemm_ksweep.log 2>&1; cat gpurun_out/r03_gemm_ksweep.log
bash tools/pmc_pass.sh a4_sq "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" tools/gemm_one.py 40 28672 37888 3584
bash tools/pmc_pass.sh a4_sq3 "SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT" tools/gemm_one.py 40 28672 37888 3584
python - <<'PY'
import json
for f in ("a4_sq","a4_sq3"):
    d=json.load(open(f"gpurun_out/pmc_{f}.json"))
    for k,v in d.items():
        if "gemm" in k: print(f, k[:60], {a:(round(b/v["dispatches"]) if a!="dispatches" else b) for a,b in v.items()})
PY
