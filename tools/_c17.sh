mkdir -p gpurun_out/r06
cd $GRAFT_REPO_ROOT
bash tools/pmc_pass.sh dattn1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_WAVES" tools/gen_flat.py 150 64 8
bash tools/pmc_pass.sh dattn2 "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM" tools/gen_flat.py 150 64 8
bash tools/pmc_pass.sh dattn3 "FETCH_SIZE" tools/gen_flat.py 150 64 8
bash tools/pmc_pass.sh dattn4 "TCC_HIT_sum TCC_MISS_sum" tools/gen_flat.py 150 64 8
python3 - <<'PY' > gpurun_out/r06/c17_pmc.txt
import json
for t in ("dattn1","dattn2","dattn3","dattn4"):
    try:
        d=json.load(open(f"gpurun_out/pmc_{t}.json"))
    except Exception as e:
        print(t, "ERR", e); continue
    for k,v in d.items():
        if 'attn_fwd128_kernel<false>' in k or 'gemm_swiglu512' in k or 'attn_merge' in k: print(t, k[:50], {a:(f"{b:.4e}" if isinstance(b,float) else b) for a,b in v.items()})
PY
cat gpurun_out/r06/c17_pmc.txt
