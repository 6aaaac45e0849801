#!/bin/bash
# K-tile schedule variants of the 4-wave fp8 tile: parity of each (a GEMM against the default's result), timing on hot / cold shapes
mkdir -p gpurun_out/r04
python3 - <<'PY' 2>&1 | tee gpurun_out/r04/mx4_sched.txt
import torch, sys
sys.path.insert(0, ".")
from spatialthinker_amd import ops
torch.manual_seed(0)
for (M, N, K) in ((300, 520, 384), (1024, 512, 128), (2048, 3584, 3584)):
    a = torch.randn(M, K, device="cuda").bfloat16(); b = torch.randn(N, K, device="cuda").bfloat16()
    aq, sa = ops.mxfp8_quantize(a); bq, sb = ops.mxfp8_quantize(b)
    ops.gemm_mxfp8_select(8); ref = ops.gemm_mxfp8_nt(aq, sa, bq, sb)
    for v in (50, 52, 56):
        ops.gemm_mxfp8_select(v)
        out = ops.gemm_mxfp8_nt(aq, sa, bq, sb)
        print(M, N, K, "schedule", v - 50, "max |diff| vs the 8-wave tile", float((out.float() - ref.float()).abs().max()), "of", float(ref.float().abs().max()))
ops.gemm_mxfp8_select(4)
PY
export MX4_SCHEDULES=6,0,2,6
timeout 300 python tools/mx4_ksweep.py 2>&1 | tee -a gpurun_out/r04/mx4_sched.txt
timeout 300 python tools/mx4_ksweep.py 16384 4608 3584 2>&1 | tee -a gpurun_out/r04/mx4_sched.txt
timeout 300 python tools/mx4_ksweep.py 16384 3584 18944 2>&1 | tee -a gpurun_out/r04/mx4_sched.txt
timeout 300 python tools/mx4_ksweep.py 16384 37888 3584 2>&1 | tee -a gpurun_out/r04/mx4_sched.txt
