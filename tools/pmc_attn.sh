cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/pmc_pass.sh attn_sq "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" tools/attn_bench.py 8 1614
bash tools/pmc_pass.sh attn_sq2 "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC" tools/attn_bench.py 8 1614
python3 - <<'PY'
import json
for t in ("attn_sq","attn_sq2"):
    d=json.load(open(f"gpurun_out/pmc_{t}.json"))
    for k,v in d.items():
        if 'attn' in k and ('128' in k): print(k[:60], {a:(f"{b:.3e}" if isinstance(b,float) else b) for a,b in v.items()})
PY
