# PMC passes over the one-pass 512-row gate/up tile (gemm_swiglu512.hip) in the mode ST_GU512_MODE selects: tools/pmc_decode_gu512.sh [rows]
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
M=${1:-512}
T=gu512_m${ST_GU512_MODE:-0}
bash tools/pmc_pass.sh ${T}_ta "TA_TA_BUSY TA_BUSY_sum TCP_TCC_READ_REQ TCP_TOTAL_READ TCP_PENDING_STALL_CYCLES GRBM_GUI_ACTIVE" tools/decode_swiglu_one.py $M 18944 3584
bash tools/pmc_pass.sh ${T}_sq "SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_WAVE_CYCLES" tools/decode_swiglu_one.py $M 18944 3584
bash tools/pmc_pass.sh ${T}_sq2 "SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY" tools/decode_swiglu_one.py $M 18944 3584
bash tools/pmc_pass.sh ${T}_l2 "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum TCP_TCC_READ_REQ_LATENCY" tools/decode_swiglu_one.py $M 18944 3584
python3 - <<PY
import json
for t in ("ta","sq","sq2","l2"):
    try: d=json.load(open(f"gpurun_out/pmc_${T}_{t}.json"))
    except Exception as e: print(t, "missing", e); continue
    for k,v in d.items():
        if 'swiglu512' in k: print("${T}", t, {a:(f"{b:.4g}" if isinstance(b,float) else b) for a,b in v.items()})
PY
