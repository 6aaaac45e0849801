mkdir -p gpurun_out/r06
python3 tools/parity_probe.py 16 --json gpurun_out/r06/bf16_yardstick.json > gpurun_out/r06/c5_parity.txt 2>&1
cp gpurun_out/r06/bf16_yardstick.json tests/golden/bf16_yardstick.json
python3 -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06/c5_smoke.txt 2>&1
timeout 1200 python3 -m pytest tests/test_gpu_model.py tests/test_gpu_depth.py -x -q -s 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -40 > gpurun_out/r06/c5_tests.txt
tail -3 gpurun_out/r06/c5_parity.txt; tail -2 gpurun_out/r06/c5_smoke.txt; tail -5 gpurun_out/r06/c5_tests.txt
