mkdir -p gpurun_out/r06
(timeout 900 python3 -m pytest tests/test_gpu_kernels.py -x -q -k "decode_attention" 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -8) > gpurun_out/r06/c18_tests.txt
cat gpurun_out/r06/c18_tests.txt
for cfg in "0 0" "1 2" "1 12" "1 13" "1 14"; do set -- $cfg
  for shape in "200 64 8" "700 64 8" "600 8 8"; do
    echo "ROWS=$1 SLOTS=$2 shape=$shape: $(ST_DECODE_ROWS=$1 ST_DECODE_ROWS_SLOTS=$2 python3 tools/gen_flat.py $shape 2>&1 | grep '^rows' | tail -1)" >> gpurun_out/r06/c18_flat.txt
  done
done
cat gpurun_out/r06/c18_flat.txt
