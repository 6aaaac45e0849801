# round 4, call 1: baseline per-kernel decode profile at 512 / 256 rows + the training-tile entries at 257..512 rows
mkdir -p gpurun_out/r04
python3 tools/decode512_probe.py 320 384 512 > gpurun_out/r04/decode512_probe.log 2>&1
bash tools/gen_flat_prof.sh 200 64 8 > gpurun_out/r04/gen_flat_512.log 2>&1
bash tools/gen_flat_prof.sh 200 32 8 > gpurun_out/r04/gen_flat_256.log 2>&1
tail -5 gpurun_out/r04/decode512_probe.log
