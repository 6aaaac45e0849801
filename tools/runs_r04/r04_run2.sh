# round 4, call 2: does a tail split pay for the 257..512-row gate/up GEMM on the asm4 tile?
mkdir -p gpurun_out/r04
python3 tools/decode512_probe2.py > gpurun_out/r04/decode512_probe2.log 2>&1
cat gpurun_out/r04/decode512_probe2.log
