mkdir -p gpurun_out/r06
(timeout 900 python3 -m pytest tests/test_gpu_kernels.py -x -q -k "decode_attention" 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -15) > gpurun_out/r06/c6_tests.txt
(timeout 600 python3 -m pytest tests/test_gpu_model.py -x -q -s -k "gradients_vs_oracle or pooled" 2>&1 | grep -E "engine vs HF|worst gradient|passed|failed|Error|assert" | head -20) > gpurun_out/r06/c6_grad.txt
for cfg in "0 0" "1 0" "1 2" "1 4"; do set -- $cfg
  for shape in "200 64 8" "700 64 8" "600 8 8" "900 4 8"; do
    echo "ROWS=$1 SLOTS=$2 shape=$shape: $(ST_DECODE_ROWS=$1 ST_DECODE_ROWS_SLOTS=$2 python3 tools/gen_flat.py $shape 2>&1 | grep '^rows' | tail -1)" >> gpurun_out/r06/c6_flat.txt
  done
done
cat gpurun_out/r06/c6_tests.txt gpurun_out/r06/c6_grad.txt gpurun_out/r06/c6_flat.txt
