# PMC passes over the decode gate/up + SwiGLU GEMM (tools/decode_swiglu_one.py M 18944 3584): memory-path counters
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
M=${1:-256}
bash tools/pmc_pass.sh gu_lat "TCP_TCC_READ_REQ_LATENCY TCP_TCC_READ_REQ TCP_TCP_LATENCY TCP_TOTAL_READ GRBM_GUI_ACTIVE" tools/decode_swiglu_one.py $M 18944 3584
bash tools/pmc_pass.sh gu_tlb "TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_HIT TCP_UTCL1_REQUEST TCP_PENDING_STALL_CYCLES TCP_TCR_TCP_STALL_CYCLES" tools/decode_swiglu_one.py $M 18944 3584
bash tools/pmc_pass.sh gu_sq "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_WAVE_CYCLES" tools/decode_swiglu_one.py $M 18944 3584
bash tools/pmc_pass.sh gu_sq2 "SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES" tools/decode_swiglu_one.py $M 18944 3584
python3 - <<'PY'
import json
for t in ("gu_lat","gu_tlb","gu_sq","gu_sq2"):
    d=json.load(open(f"gpurun_out/pmc_{t}.json"))
    for k,v in d.items():
        if 'gemm_tile' in k: print(t, {a:(f"{b:.4g}" if isinstance(b,float) else b) for a,b in v.items()})
PY
